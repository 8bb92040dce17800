"""tools/train_ddp.py: the process-per-GPU launcher around an UNCHANGED single-device training script.  CPU, gloo, 2 ranks;
the stand-in script follows the conventions of the reference's scripts/train_diffusion.py (positional config / output
directory, --seed, a shuffling DataLoader, imports through scene_synthesis.networks)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import argparse, json, os, sys
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Dataset
import scene_synthesis.networks as nets            # must resolve to diffuscene_amd.networks

class Rooms(Dataset):
    def __len__(self): return 22
    def __getitem__(self, i): return i

p = argparse.ArgumentParser()
p.add_argument("config_file"); p.add_argument("output_directory"); p.add_argument("--seed", type=int, default=27)
a = p.parse_args()
loader = DataLoader(Rooms(), batch_size=4, shuffle=True, num_workers=0)
passes = [[int(v) for b in loader for v in b] for _ in range(2)]
rank = dist.get_rank() if dist.is_initialized() else 0
rec = {"rank": rank, "world": dist.get_world_size() if dist.is_initialized() else 1, "passes": passes,
       "out": a.output_directory, "nets": nets.__name__, "has_build_network": hasattr(nets, "build_network"),
       "visible": os.environ.get("HIP_VISIBLE_DEVICES")}
os.makedirs(a.output_directory, exist_ok=True)
with open(os.path.join(os.environ["DSC_TEST_DIR"], "rank%d.json" % rank), "w") as f:
    json.dump(rec, f)
'''


def test_launcher_shards_loader_and_redirects_rank_outputs(tmp_path):
    script = tmp_path / "fake_train.py"
    script.write_text(SCRIPT)
    out = tmp_path / "out"
    env = dict(os.environ, DSC_TEST_DIR=str(tmp_path), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_ddp.py"), "--reference", str(tmp_path), "--gpus", "2",
                        "--script", str(script), "--", "cfg.yaml", str(out), "--seed", "5"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    recs = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(2)]
    assert [x["rank"] for x in recs] == [0, 1] and all(x["world"] == 2 for x in recs)
    assert all(x["nets"] == "diffuscene_amd.networks" and x["has_build_network"] for x in recs)
    assert [x["visible"] for x in recs] == ["0", "1"]                 # cuda:0 of the script = the rank's own GPU
    assert recs[0]["out"] == str(out) and recs[1]["out"] != str(out)   # only rank 0 writes where the user asked
    for e in range(2):
        a, b = recs[0]["passes"][e], recs[1]["passes"][e]
        assert len(a) == len(b) == 11 and sorted(a + b) == list(range(22))   # disjoint shards that cover the dataset
    assert recs[0]["passes"][0] != recs[0]["passes"][1]               # a new permutation on every pass


def _json_objects(text):
    """Every JSON object in the ranks' shared stdout (two ranks may land on one line)."""
    dec, out = json.JSONDecoder(), []
    for line in text.splitlines():
        i = line.find("{")
        while i >= 0:
            try:
                obj, end = dec.raw_decode(line, i)
            except ValueError:
                break
            out.append(obj)
            i = line.find("{", end)
    return out


def test_bench_self_spawns_one_rank_per_gpu():
    """`python bench.py --gpus N` from a plain shell (how the driver calls it) re-executes itself under torch.distributed.run
    with N ranks on 127.0.0.1; DSC_BENCH_DRYRUN stops every rank before it touches a GPU."""
    env = dict(os.environ, DSC_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--scaling", "strong"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    recs = sorted(_json_objects(r.stdout), key=lambda d: d["rank"])
    assert [d["rank"] for d in recs] == [0, 1] and all(d["world"] == 2 and d["gpus"] == 2 for d in recs)
    assert [d["local_rank"] for d in recs] == [0, 1] and all(d["master"] == "127.0.0.1" and d["scaling"] == "strong" for d in recs)
    assert all(d["batch_per_rank"] == 128 and d["ddp_flush"] == "single" and d["ipc_legacy"] == "0" for d in recs)
    # the driver's 8-GPU line, cold: `python bench.py --gpus 8` (strong scaling, another flush schedule through the environment) --
    # eight ranks rendezvous on 127.0.0.1, every rank sees its own local rank, the 32-scene shard and the flush mode
    r8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--scaling", "strong"],
                        env=dict(env, DSC_DDP_FLUSH="thirds"), capture_output=True, text=True, timeout=600)
    assert r8.returncode == 0, r8.stdout + r8.stderr
    recs8 = sorted(_json_objects(r8.stdout), key=lambda d: d["rank"])
    assert [d["local_rank"] for d in recs8] == list(range(8)) and all(d["world"] == 8 and d["gpus"] == 8 for d in recs8)
    assert all(d["batch_per_rank"] == 32 and d["ddp_flush"] == "thirds" and d["scaling"] == "strong" for d in recs8)
    # under torchrun with a mismatching --gpus the script refuses instead of running a wrong configuration
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env2, capture_output=True, text=True,
                        timeout=120)
    assert r2.returncode != 0


def test_bench_side_line_child_contains_failures(monkeypatch):
    """bench.side_line_child: whatever happens in a side line's child process (GPU fault -> abort, exception, hang) becomes an
    {"error": ...} entry of the parent's record; a clean child contributes the JSON line it printed (after any library chatter)."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench

    class R:
        def __init__(self, rc, out, err=""):
            self.returncode, self.stdout, self.stderr = rc, out, err
    seen = []

    def fake_run(cmd, env=None, capture_output=None, text=None, timeout=None):
        seen.append((cmd, env))
        return results.pop(0)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("RANK", "0")
    results = [R(0, 'RCCL version ...\n{"steps_per_s": 101.5, "workload": "x"}\ntrailing chatter\n')]
    assert bench.side_line_child("arrange") == {"steps_per_s": 101.5, "workload": "x"}
    cmd, env = seen[-1]
    assert cmd[-2:] == ["--side-line", "arrange"] and "RANK" not in env              # a single-GPU child, whatever launched the parent
    results = [R(-6, "", "Memory access fault by GPU node-2 (Agent handle: 0x1) on address 0x7000. Reason: Unknown.\n")]
    out = bench.side_line_child("living80:f32")
    assert set(out) == {"error"} and "exit code -6" in out["error"] and "Memory access fault" in out["error"]
    results = [R(0, "no json here\n")]
    assert "error" in bench.side_line_child("text")

    def hang(cmd, env=None, capture_output=None, text=None, timeout=None):
        raise subprocess.TimeoutExpired(cmd, timeout)
    monkeypatch.setattr(subprocess, "run", hang)
    assert bench.side_line_child("complete", timeout=7) == {"error": "timed out after 7 s"}
