"""Synthetic workloads of the five BASELINE.json configurations: network keyword sets of the shipped YAMLs (scaled to the benchmark's
object counts), a synthetic ``dataset_stats.txt`` and scene batches with the value distribution of the real encoders (SURVEY.md 8d;
reference datasets/threed_front_dataset.py:377-382,500-507,906-921).  Used by bench.py and the measurement tools to build the PRODUCT's
model and inputs; oracle/weights.py re-exports the same objects so that tests and the CPU baseline see the identical workload."""
import numpy as np
import torch


def synth_scene_batch(B, N, class_dim, objfeat_dim=32, seed=0):
    """(B, N, C) fp32, C = 3+3+2+class_dim+objfeat_dim, channel order of
    diffusion_scene_layout_ddpm.py:148-154."""
    rng = np.random.RandomState(seed)
    C = 8 + class_dim + objfeat_dim
    x = np.zeros((B, N, C), dtype=np.float32)
    for b in range(B):
        n_obj = rng.randint(min(3, N), N + 1)
        x[b, :n_obj, 0:6] = rng.uniform(-1, 1, size=(n_obj, 6))
        th = rng.uniform(-np.pi, np.pi, size=n_obj)
        x[b, :n_obj, 6] = np.cos(th)
        x[b, :n_obj, 7] = np.sin(th)
        cls = -np.ones((N, class_dim), dtype=np.float32)
        cls[np.arange(n_obj), rng.randint(0, class_dim - 1, size=n_obj)] = 1.0
        cls[n_obj:, class_dim - 1] = 1.0
        x[b, :, 8:8 + class_dim] = cls
        if objfeat_dim:
            x[b, :n_obj, 8 + class_dim:] = rng.uniform(-1, 1, size=(n_obj, objfeat_dim))
    return torch.from_numpy(x)


DATASET_STATS = {  # synthetic dataset_stats.txt (diffusion_ddpm.py:137-151)
    "bounds_translations": [-2.76, 0.045, -2.75, 2.78, 3.62, 2.82],
    "bounds_sizes": [0.04, 0.02, 0.01, 2.87, 1.77, 1.70],
    "bounds_angles": [-3.1416, 3.1416],
}

UNCOND_BEDROOM = dict(dim=512, dim_mults=[1, 1, 1, 1], channels=62, objectness_dim=0, objfeat_dim=32,
                      class_dim=22, angle_dim=2, context_dim=0, instanclass_dim=128, seperate_all=True)
UNCOND_LIVING = dict(dim=512, dim_mults=[1, 1, 1, 1], channels=65, objectness_dim=0, objfeat_dim=32,
                     class_dim=25, angle_dim=2, context_dim=0, instanclass_dim=128, seperate_all=True)
TEXT_BEDROOM = dict(dim=512, dim_mults=[1, 1, 1, 1], channels=62, objectness_dim=0, class_dim=22,
                    angle_dim=2, objfeat_dim=32, self_condition=True, context_dim=0, instanclass_dim=128,
                    seperate_all=True, merge_bbox=True, modulate_time_context_instanclass=True,
                    text_condition=True, text_dim=512)
REARRANGE_LIVING = dict(dim=512, dim_mults=[1, 1, 1, 1], channels=5, objectness_dim=0, class_dim=25,
                        angle_dim=2, objfeat_dim=32, self_condition=True, context_dim=0,
                        instanclass_dim=512, modulate_time_context_instanclass=True)
