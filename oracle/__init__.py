"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the DiffuScene DDPM hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the CPU baseline being timed.  The product
package (``diffuscene_amd``) never imports this package and raises when the HIP
library is missing instead of falling back to anything here.

Contents
--------
ref_torch.py   plain PyTorch fp32 CPU restatement of the reference algorithm
               (Unet1D forward, GaussianDiffusion schedule / q_sample / p_sample /
               p_losses / loops, 3-D IoU), every function citing the reference
               file:line it follows.
weights.py     deterministic synthetic weights / inputs shared by oracle, golden
               generator and tests (checkpoints are not available offline).
ref_loader.py  imports the REAL reference modules from /root/reference (only
               exists in the build container, never on the GPU box).
make_golden.py runs the real reference and writes tests/golden/*.npz; the
               restatement is pinned against those vectors by tests/test_oracle.py.

Parity status: PINNED -- the restatement is checked against outputs of the
reference's own modules (denoise_net.py, diffusion_ddpm.py, loss.py) executed in
the build container; the generating script and the vectors are committed.
"""
