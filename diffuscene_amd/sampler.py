"""hipGraph-captured reverse diffusion loop (reference p_sample_loop, diffusion_ddpm.py:355-371).

The reference launches ~800 kernels per step from Python, 8e5 launches per sample.  Here ONE
reverse step -- the denoiser launch plan (~140 kernels), the noise draw and the fused posterior step -- is
captured once into a hipGraph whose only state is device-resident (x_t, the int64 timestep vector, the
conditioning buffers of the plan) and replayed T times; the timestep is decremented by a kernel inside the graph.
RNG draw order is the reference's: x_T first, then one draw per step (also at t == 0).
"""
import torch

from . import ops
from .networks.denoise_net import Unet1D

_MEAN = {"eps": ops.MEAN_EPS, "x0": ops.MEAN_X0, "v": ops.MEAN_V}


class NoiseReplay:
    """noise_fn (protocol of diffusion_ddpm.py:345,355-356) that replays a pre-generated device tensor
    ``buffer[i]`` for the i-th draw.  Usable eagerly and inside the captured graph (parity tests inject the
    reference's noise this way)."""

    def __init__(self, buffer, partial_buffer=None):
        self.buffer = buffer                      # (T+1, B, N, C): x_T, then the p_sample draw of every step
        self.partial_buffer = partial_buffer      # (T, B, P, C): completion only, the draw that re-noises the given objects
        self.i = 0
        self.ip = 0

    def __call__(self, size=None, dtype=None, device=None):
        if self.partial_buffer is not None and tuple(size) == tuple(self.partial_buffer.shape[1:]) \
                and tuple(size) != tuple(self.buffer.shape[1:]):
            n = self.partial_buffer[self.ip]
            self.ip += 1
            return n
        n = self.buffer[self.i]
        self.i += 1
        assert tuple(n.shape) == tuple(size), (tuple(n.shape), tuple(size))
        return n


def _chains_for(B):
    """Independent sub-batch chains per captured step (env DSC_CHAINS, default 1).  Scenes are independent, so the batch
    can run as several dependency chains on separate streams inside the graph.  Measured on MI355X (B=256, N=80,
    profiles/r02_chain_sweep_*.txt): with the round-1 GEMMs two 128-scene chains gained 3 % (12.07 vs 12.44 ms; starting the
    second chain 15-110 us late so that its K loops run under the first chain's epilogues lost on every offset); with the
    interleaved LDS-DMA GEMMs one chain is best (10.73 ms vs 10.97 ms for two, 12.1 ms for four) -- a single 256-scene chain
    already fills the CUs and leaves no launch gaps -- so it stays opt-in (useful when the per-GPU batch is far above 256)."""
    import os
    n = int(os.environ.get("DSC_CHAINS", "1"))
    return n if (n > 1 and B % n == 0 and B // n >= 64) else 1


class _StepGraph:
    def __init__(self, diff, model, shape, device, condition, condition_cross, clip_denoised, replay=False,
                 partial_shape=None):
        B, N, C = shape
        self.shape = shape
        eng = model.engine(device)
        use_table = diff.num_timesteps <= eng.time_table.shape[0]
        nch = _chains_for(B)
        Bc = B // nch
        self.plans = [eng.prepare(Bc, N, None if condition is None else condition[i * Bc:(i + 1) * Bc],
                                  None if condition_cross is None else condition_cross[i * Bc:(i + 1) * Bc],
                                  time_table=use_table, slot=i) for i in range(nch)]
        self.plan = self.plans[0]
        self.side = [torch.cuda.Stream(device=device) for _ in range(nch - 1)]
        tb = diff.tables(device)
        ca, cb = diff._coeffs(tb)
        self.x = torch.empty(shape, device=device, dtype=torch.float32)
        self.t = torch.zeros((B,), device=device, dtype=torch.int64)
        mean_type = _MEAN[diff.model_mean_type]
        sigma = diff._sigma(tb)
        plan = self.plan
        self.replay = replay
        self.noise_buf = None                                   # (T+1, B, N, C) when replaying
        self.draw = torch.zeros((1,), device=device, dtype=torch.int64)
        # scene completion (p_sample_loop_complete, diffusion_ddpm.py:461-466): the given objects are re-noised and
        # written over the first P rows of x_t at every step, BEFORE the model call
        self.partial = torch.zeros(partial_shape, device=device) if partial_shape is not None else None
        self.pnoise_buf = None
        self.pdraw = torch.zeros((1,), device=device, dtype=torch.int64)

        xv = self.x.view(B * N, C)
        self.model_out = torch.empty(shape, device=device, dtype=torch.float32) if nch > 1 else None

        def run_chain(i):
            p = self.plans[i]
            p.x_in.copy_(xv[i * Bc * N:(i + 1) * Bc * N])
            p.t_in.copy_(self.t[i * Bc:(i + 1) * Bc])
            p.run()
            if nch > 1:
                self.model_out.view(B * N, C)[i * Bc * N:(i + 1) * Bc * N].copy_(p.out)

        def step():
            if self.partial is not None:
                if self.replay:
                    pn = self.pnoise_buf.index_select(0, self.pdraw)[0]
                    ops.add_scalar_i64(self.pdraw, 1)
                else:
                    pn = torch.randn(partial_shape, dtype=torch.float, device=device)
                ops.complete_overwrite(self.x, self.partial, pn, self.t, tb["sqrt_alphas_cumprod"],
                                       tb["sqrt_one_minus_alphas_cumprod"])
            cur = torch.cuda.current_stream(device)
            for st in self.side:
                st.wait_stream(cur)
            run_chain(0)
            for i, st in enumerate(self.side):
                with torch.cuda.stream(st):
                    run_chain(i + 1)
            for st in self.side:
                cur.wait_stream(st)
            if self.replay:
                noise = self.noise_buf.index_select(0, self.draw)[0]
                ops.add_scalar_i64(self.draw, 1)
            else:
                noise = torch.randn(shape, dtype=torch.float, device=device)
            ops.p_sample(self.x, self.model_out if nch > 1 else plan.out.view(B, N, C), noise, self.t, ca, cb,
                         tb["posterior_mean_coef1"],
                         tb["posterior_mean_coef2"], sigma, mean_type, clip_denoised, out=self.x)
            ops.add_scalar_i64(self.t, -1)

        # warm-up on a side stream (loads every code object, sizes the allocator), then capture.  Building the graph must not cost the
        # caller random numbers: the loop is the default path behind the reference's call sites (round 6), and a run seeded with
        # torch.manual_seed has to draw what the eager loop draws whether or not this call had to capture first -- the device
        # generator's state is put back afterwards (the warm-up step and normal_() below draw from it).
        rng_state = torch.cuda.get_rng_state(device)
        self.x.normal_()
        self.t.fill_(1)
        if replay:
            self.noise_buf = torch.zeros((diff.num_timesteps + 1,) + tuple(shape), device=device)
            if partial_shape is not None:
                self.pnoise_buf = torch.zeros((diff.num_timesteps,) + tuple(partial_shape), device=device)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            step()
        torch.cuda.set_rng_state(rng_state, device)

    def check_current(self):
        """The captured launches hold raw pointers into the parameter storages of capture time.  If the parameters have moved since
        (the first training step re-homes them into flat storage, module.to() ...) the graph is stale: refuse to replay it.  (The
        plans keep the old storages alive, so a stale replay could not fault -- it would silently sample from old weights.)"""
        for p in self.plans:
            p.eng.params_moved()
            p.check_current()

    def replay_steps(self, n=1):
        """n captured reverse steps on the current stream (x, t and the conditioning buffers are the graph's own state)."""
        self.check_current()
        for _ in range(n):
            self.graph.replay()

    def run(self, x_T, total_steps, noise_buffer=None, partial=None, partial_noise=None):
        self.check_current()
        self.x.copy_(x_T)
        self.t.fill_(total_steps - 1)
        if self.partial is not None:
            self.partial.copy_(partial)
        if self.replay:
            self.noise_buf[:noise_buffer.shape[0]].copy_(noise_buffer)
            self.draw.fill_(1)                                  # draw 0 was x_T
            if self.partial is not None:
                self.pnoise_buf[:partial_noise.shape[0]].copy_(partial_noise)
                self.pdraw.fill_(0)
        for _ in range(total_steps):
            self.graph.replay()
        out = self.x.clone()
        # the in-graph timestep now holds -1: park it on a valid row, so that one replay too many (a caller driving `graph` directly)
        # still indexes the schedule tables in range (the kernels clamp and count it either way: dsc_device_error_count)
        self.t.fill_(0)
        return out


def graph_sample_loop(diff, denoise_fn, shape, device, condition, condition_cross, clip_denoised, total_steps,
                      noise_fn=torch.randn, partial_boxes=None):
    model = getattr(getattr(denoise_fn, "__self__", None), "model", None)
    if not isinstance(model, Unet1D):
        raise RuntimeError("graph sampling needs DiffusionPoint._denoise over a diffuscene_amd Unet1D")
    device = torch.device(device)
    with torch.no_grad():
        replay = isinstance(noise_fn, NoiseReplay)
        pshape = None if partial_boxes is None else tuple(partial_boxes.shape)
        key = (id(model), tuple(shape), str(device), bool(clip_denoised), diff.model_mean_type, replay, pshape,
               None if condition is None else (tuple(condition.shape), condition.stride(0) == 0),
               None if condition_cross is None else tuple(condition_cross.shape))
        g = diff._graphs.get(key)
        eng = model.engine(device)
        eng.params_moved()              # parameters re-homed since the capture (first training step, .to()): the engine drops its plans
        if g is None or g.plan is not eng.plans.get(_plan_key(g)):
            g = _StepGraph(diff, model, tuple(shape), device, condition, condition_cross, clip_denoised, replay, pshape)
            diff._graphs = {key: g}           # one live graph per diffusion object
        else:
            nch = len(g.plans)
            Bc = shape[0] // nch
            for i in range(nch):                                     # refresh weights + conditioning buffers
                eng.prepare(Bc, shape[1], None if condition is None else condition[i * Bc:(i + 1) * Bc],
                            None if condition_cross is None else condition_cross[i * Bc:(i + 1) * Bc],
                            time_table=g.plan.time_table, slot=i)
        if replay:
            out = g.run(noise_fn.buffer[0], total_steps, noise_fn.buffer, partial_boxes, noise_fn.partial_buffer)
        else:
            x_T = torch.randn(shape, dtype=torch.float, device=device)
            out = g.run(x_T, total_steps, partial=partial_boxes)
        if partial_boxes is not None:
            out[:, :partial_boxes.shape[1], :] = partial_boxes          # clean objects restored after the last step (:471-473)
        from ._lib import check_indices
        check_indices("graph_sample_loop")     # DSC_CHECK_INDICES=1 (debugging; synchronises)
        return out


def _plan_key(g):
    p = g.plan
    return (p.B, p.N, p.ctx_mode, 0 if p.ctx_in is None else p.ctx_in.shape[1], p.L,
            0 if p.cross_in is None else p.cross_in.shape[1], p.time_table)        # slot 0 carries no suffix
