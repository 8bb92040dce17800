#!/bin/bash
# Is the fp32-MFMA GEMM power-capped?  Sample rocm-smi power / clocks while a GEMM variant loops for a few seconds.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -40
python - <<'PY' &
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from diffuscene_amd import _lib, ops
lib = C.CDLL("tools/libgemm_tune.so")
lib.tune_launch.argtypes = [C.c_int, C.c_int, C.POINTER(_lib.GemmArgs), C.c_void_p]
dev = torch.device("cuda:0")
M = 20480
a = torch.randn(M, 512, device=dev); w = torch.randn(512, 512, device=dev) * 0.05; y = torch.zeros(M, 512, device=dev)
b = torch.randn(512, device=dev)
g = ops.make_gemm_args(a, w, y, b, None, None)
s = ops.stream_ptr()
for v in (24, 23):
    t0 = time.time()
    n = 0
    while time.time() - t0 < 6.0:
        for _ in range(200):
            lib.tune_launch(v, 0, C.byref(g), s)
        torch.cuda.synchronize()
        n += 200
    print("variant %d: %.1f us per launch over %.1f s" % (v, (time.time() - t0) / n * 1e6, time.time() - t0), flush=True)
PY
sleep 2.5
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo
  sleep 1.2
done
wait
