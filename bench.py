#!/usr/bin/env python
"""bench.py -- DecNet hot-path throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic stereo pairs that are
already resident in HBM (BASELINE config 2: batch 8, 960x540 -> padded 972x540,
max_disp 192 -> 216 as the reference rounds it, demo.py:153):

    stage 0   cost volume -> 7x Conv3d(216,216,3^3)+BN+ReLU (+residual) -> Conv3d(216,1)+BN
              -> soft-argmax                                    [8,216,20,36], D=8
    stage 1-3 fused SpaMat+SpaVar (masked correlation -> softmax expectation + variance)
              [8,72,60,108] D=24 / [8,24,180,324] D=72 / [8,8,540,972] D=216
              (independent of stage 0's result: enqueued on a second HIP stream beside it)
    N>1       one RCCL all_gather of the per-rank disparity maps (the reference's
              DataParallel gather, eval.py:146) -- pairs are sharded, weak scaling; the gather
              of step k overlaps step k+1.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without WORLD_SIZE: starts N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --config 5 --gpus 4                   (training share: SpaMat fwd+bwd + gradient all-reduce)

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3     # dense fp32 matrix peak (v_mfma_f32_*_f32)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 matrix peak (MI355X_MICROARCH.md)

# BASELINE.json configs that fit one process per GPU: (name, padded H, padded W, rounded max_disp, pairs per GPU)
CONFIGS = {
    2: ("config 2: synthetic 960x540 (padded 972x540), max_disp 192->216", 540, 972, 216, 8),
    3: ("config 3: KITTI 1242x375 (padded 1242x378), max_disp 192->216, batch 32 over 8 GPUs", 378, 1242, 216, 4),
    4: ("config 4: Middlebury half-res 1500x1000 (padded 1512x1026), max_disp 256->270", 1026, 1512, 270, 1),
    5: ("config 5: Sceneflow training step share (SpaMat forward+backward, stages 1-3), 972x540, max_disp 216, "
        "batch 16 over 4 GPUs", 540, 972, 216, 4),
}
N_PARAMS = 13_190_000        # parameters of the shipped base_channels=8 network: 52.7 MB of fp32 gradients
CONFIG_NAME, PAD_H, PAD_W, MAX_DISP, DEFAULT_B = CONFIGS[2]


_JSON_OUT = None       # the process's real stdout when a process group is up (see main)


def stage_shapes(h, w, md):
    """(C, H, W, D) for stage 0..3 of the shipped 4-stage / scale-3 network."""
    return [(216, h // 27, w // 27, md // 27), (72, h // 9, w // 9, md // 9), (24, h // 3, w // 3, md // 3),
            (8, h, w, md)]


STAGES = stage_shapes(PAD_H, PAD_W, MAX_DISP)


def set_config(n):
    global CONFIG_NAME, PAD_H, PAD_W, MAX_DISP, DEFAULT_B, STAGES
    CONFIG_NAME, PAD_H, PAD_W, MAX_DISP, DEFAULT_B = CONFIGS[n]
    STAGES = stage_shapes(PAD_H, PAD_W, MAX_DISP)


def make_inputs(B, dev, density, seed):
    """Op-level synthetic tensors: features relu(N(0,1)), Bernoulli masks."""
    feats, masks = [], []
    for s, (C, H, W, D) in enumerate(STAGES):
        g = torch.Generator(device=dev).manual_seed(seed + 17 + s)
        L = torch.relu(torch.randn(B, C, H, W, device=dev, generator=g))
        R = torch.relu(torch.randn(B, C, H, W, device=dev, generator=g))
        feats.append((L, R))
        g = torch.Generator(device=dev).manual_seed(seed + 170 + s)
        if density >= 1.0:
            rm = torch.ones(B, H, W, device=dev)
            tm = torch.ones(B, H, W, device=dev)
        else:
            rm = (torch.rand(B, H, W, device=dev, generator=g) < density).float()
            tm = (torch.rand(B, H, W, device=dev, generator=g) < density).float()
        masks.append((rm, tm))
    return feats, masks


def make_regularizer(C, dev, seed=17, cost_func="cor"):
    """Random-init CostRegNetNoDown (conv init as SparseDenseNetRefinementMask.py:248-250)."""
    import decnet_amd
    reg = decnet_amd.CostRegNetNoDown(in_channels=C, base_channels=2 * C, cost_func=cost_func)
    g = torch.Generator().manual_seed(seed)
    if cost_func == "cat":
        reg.conv_pre.weight.data.normal_(0, math.sqrt(2.0 / C), generator=g)
    for u in reg.units():
        co = u.conv.weight.shape[0]
        u.conv.weight.data.normal_(0, math.sqrt(2.0 / (27 * co)), generator=g)
        u.bn.weight.data = torch.rand(co, generator=g) * 0.5 + 0.75
        u.bn.bias.data = torch.randn(co, generator=g) * 0.1
        u.bn.running_mean.data = torch.randn(co, generator=g) * 0.1
        u.bn.running_var.data = torch.rand(co, generator=g) * 0.5 + 0.75
    return reg.to(dev).eval()


class HotPath:
    def __init__(self, B, dev, density, world):
        import decnet_amd
        from decnet_amd import dist as decnet_dist
        self.decnet, self.dist = decnet_amd, decnet_dist
        self.B, self.dev, self.world = B, dev, world
        self.coll = world > 1 or decnet_dist.force_collective()     # the per-step all-gather runs (RCCL)
        self.feats, self.masks = make_inputs(B, dev, density, seed=1000 * (int(os.environ.get("RANK", 0)) + 1))
        self.reg = make_regularizer(STAGES[0][0], dev)
        self.stage0 = decnet_amd.Stage0(self.reg)
        self.outs = [tuple(torch.empty(B, H, W, device=dev) for _ in range(4)) for (_, H, W, _) in STAGES[1:]]
        # N > 1: the all-gather of step k overlaps the kernels of step k+1, so the stage-3 outputs and
        # the receive buffers are double buffered
        self.outs3 = [self.outs[2], tuple(torch.empty_like(t) for t in self.outs[2])]
        self.gbuf, self.pending, self.k = [None, None], [None, None], 0
        self.side = None
        self.ev = None
        self.gstream = None                             # the stream the all-gather is issued from (round 4)
        self.overlap = True                             # --no-overlap: everything in order on one stream

    def drain(self):
        """Wait for the all-gathers still in flight (end of a timed region)."""
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None

    def step(self, events=None):
        """One pass.  SpaMat/SpaVar of stages 1-3 depend on the features and masks only, not on stage 0's
        result, so they run on a second HIP stream beside the stage-0 convolution stack (they fill its
        kernel tails: ~3 %).  With `events` (the untimed breakdown steps) everything runs in order on one
        stream so that the per-stage times are clean."""
        d = self.decnet
        cur = torch.cuda.current_stream()
        overlap = events is None and self.overlap
        if overlap:
            if self.side is None:
                self.side = torch.cuda.Stream()
            self.side.wait_stream(cur)

        def stage0():
            L, R = self.feats[0]
            if events is not None:
                events["s0_beg"].record()
            p = self.stage0(L, R, STAGES[0][3])
            if events is not None:
                events["s0_end"].record()
            return p

        def sparse_stages():
            for s in (3, 2, 1):
                (L, R), (rm, tm) = self.feats[s], self.masks[s]
                if events is not None and s == 3:
                    events["s3_beg"].record()
                par = self.k & 1 if self.coll else 0
                if s == 3 and self.pending[par] is not None:  # the gather that read this buffer set two steps ago
                    self.pending[par].wait()
                    self.pending[par] = None
                d.spamatvar_forward(L, R, rm, tm, STAGES[s][3], out=self.outs3[par] if s == 3 else self.outs[s - 1])
                if events is not None and s == 3:
                    events["s3_end"].record()

        gather_own = self.coll
        ev3 = None
        if overlap:
            with torch.cuda.stream(self.side):
                sparse_stages()
                if gather_own:
                    ev3 = torch.cuda.Event()
                    ev3.record()                                     # stage-3 disparity of this step is complete
            pred0 = stage0()
            if not gather_own:
                cur.wait_stream(self.side)
        else:
            pred0 = stage0()
            sparse_stages()
        disp = self.outs3[self.k & 1 if self.coll else 0][0]
        if self.coll:            # one RCCL all-gather of the per-rank disparity maps, not waited for here
            par = self.k & 1
            if gather_own:
                # issued from its own stream, ordered behind the stage-3 kernels by ONE event: RCCL's stream then follows
                # the side stream only, and the main stream (stage 0 of this and the next step) is never joined to it
                if self.gstream is None:
                    self.gstream = torch.cuda.Stream()
                if ev3 is None:
                    ev3 = torch.cuda.Event()
                    ev3.record()
                self.gstream.wait_event(ev3)
                with torch.cuda.stream(self.gstream):
                    disp, self.pending[par] = self.dist.gather_disparity(disp, n_pairs=self.world * self.B,
                                                                         out=self.gbuf[par], async_op=True)
            else:
                disp, self.pending[par] = self.dist.gather_disparity(disp, n_pairs=self.world * self.B,
                                                                     out=self.gbuf[par], async_op=True)
            self.gbuf[par] = disp
            self.k += 1
        return pred0, disp


def time_kernel(fn, iters, warm=10):
    for _ in range(warm):
        fn()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return beg.elapsed_time(end) / iters          # ms per call


def valu_floor(n_wave_tiles, n_mfma_per_tile=30, mfma_issue_cycles=8):
    """The instruction-issue floor of the dense stage-3 cost-volume pass, measured live: tools/ubench/
    softmax_rate.bin (built by __graft_entry__.build(); run as a child process) times nothing but the VALU work
    of the kernel's softmax passes -- 60 candidates per lane, the 'per-tile moments' formulation the kernel
    uses, 4 waves per SIMD on every CU -- and the loop overhead of the microbenchmark itself, which is
    subtracted.  The MFMAs of a wave-tile are priced by the cycles they hold the SIMD's vector issue port at the
    nominal 2.4 GHz: 8 of its 16 cycles for v_mfma_f32_16x16x32_bf16 (the default dense path: bf16x3 cost tiles, the
    matrix pipe works beside the other waves' VALU passes), all 32 for v_mfma_f32_16x16x4_f32
    (DECNET_SPAMAT_DENSE=fp32: fp32 MFMA runs on the FP32 lanes the VALU passes use, the two add -- ablation builds,
    DESIGN.md).  floor = wave-tiles per SIMD x (softmax time + MFMA issue time) per wave-tile."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench", "softmax_rate.bin")
    if not os.path.exists(exe):
        return None
    try:
        txt = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    except Exception:                                   # noqa: BLE001 -- an extra object, never the bench line
        return None
    ms = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^(.+?)\s+([0-9.]+) ms", txt, re.M)}
    if "per-tile moments" not in ms or "inputs only (loop overhead)" not in ms:
        return None
    ub_tiles_per_simd = 256 * 4 * 4 * 512 / 1024.0      # the microbenchmark's grid: blocks x waves x iterations / SIMDs
    soft_us = 1e3 * (ms["per-tile moments"] - ms["inputs only (loop overhead)"]) / ub_tiles_per_simd
    mfma_us = n_mfma_per_tile * mfma_issue_cycles / 2400.0
    per_simd = n_wave_tiles / 1024.0
    return {"softmax_us_per_wave_tile": soft_us, "mfma_us_per_wave_tile": mfma_us,
            "floor_ms": per_simd * (soft_us + mfma_us) * 1e-3, "floor_softmax_only_ms": per_simd * soft_us * 1e-3,
            "ubench_ms": ms}


def bwd_valu_floor(n_tiles):
    """The instruction-issue floor of the dense-row SpaMat backward (spamat_bwd_rowb), measured live like valu_floor():
    tools/ubench/bwd_tile_rate.bin times one 16 x 16 band tile's own arithmetic -- per candidate fma, exp2, the weight
    e (d - out), its split into three bf16 terms; six v_mfma_f32_16x16x32_bf16 per tile -- at 4 waves per SIMD on every
    CU, minus the microbenchmark's loop overhead.  floor = tiles per SIMD x that; what the kernel spends beyond it is
    addressing, LDS traffic, the per-left-tile staging and the barriers."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench", "bwd_tile_rate.bin")
    if not os.path.exists(exe):
        return None
    try:
        txt = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    except Exception:                                   # noqa: BLE001
        return None
    ms = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^(.+?)\s+([0-9.]+) ms", txt, re.M)}
    need = ("tile: vector work + six MFMAs", "inputs only (loop overhead)")
    if any(k not in ms for k in need):
        return None
    ub_tiles_per_simd = 256 * 4 * 4 * 2048 / 1024.0     # the microbenchmark's grid: blocks x waves x iterations / SIMDs
    tile_us = 1e3 * (ms[need[0]] - ms[need[1]]) / ub_tiles_per_simd
    return {"tile_us_per_simd": tile_us, "tiles": n_tiles, "floor_ms": n_tiles / 1024.0 * tile_us * 1e-3, "ubench_ms": ms}


def cpu_baseline(seed=17, budget_s=12.0, max_pairs=16):
    """The CPU checker (oracle/: C+OpenMP restatement of the CUDA kernels, torch-CPU stage 0)
    timed on this box's host cores, pair after pair of the same workload (the first one is a
    warm-up: library load, thread pool) until ~budget_s seconds of CPU work or max_pairs."""
    import oracle
    from oracle import stage0 as o0
    oracle.build()
    cores = oracle.num_threads()
    torch.set_num_threads(cores)
    feats, masks = [], []
    for s, (C, H, W, D) in enumerate(STAGES):
        g = torch.Generator().manual_seed(seed + s)
        feats.append((torch.relu(torch.randn(1, C, H, W, generator=g)),
                      torch.relu(torch.randn(1, C, H, W, generator=g))))
        masks.append((torch.ones(1, H, W), torch.ones(1, H, W)))
    params = o0.random_params(STAGES[0][0], 3)

    def one_pair():
        t0 = time.time()
        with torch.no_grad():
            o0.stage0_forward(feats[0][0], feats[0][1], params, STAGES[0][3])
        t1 = time.time()
        for s in (1, 2, 3):
            (L, R), (rm, tm) = feats[s], masks[s]
            o, _, _ = oracle.spamat_forward(L, R, rm, tm, STAGES[s][3])
            oracle.spavar_forward(L, R, rm, tm, o, STAGES[s][3])
        return t1 - t0, time.time() - t1

    warm = sum(one_pair())
    t_s0 = t_sp = 0.0
    n = 0
    while n < max_pairs and (n == 0 or warm + t_s0 + t_sp + (t_s0 + t_sp) / n < budget_s):
        a, b = one_pair()
        t_s0, t_sp, n = t_s0 + a, t_sp + b, n + 1
    total = t_s0 + t_sp
    res = {"value": n / total, "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": "%d pairs 972x540 max_disp 216 one after another (after 1 warm-up pair), mask density 1.0: "
                     "torch-CPU stage 0 %.2f s + C/OpenMP SpaMat+SpaVar stages 1-3 %.2f s per pair"
                     % (n, t_s0 / n, t_sp / n)}
    # "also a 1-thread number": ONE WHOLE pair, every stage measured in full (~5 s; round 3 extrapolated
    # stage 3 from a sixth of its rows).  A small budget (the -m gpu contract test) keeps the sixth-of-the-rows sample.
    try:
        oracle.set_num_threads(1)
        torch.set_num_threads(1)
        t0 = time.time()
        with torch.no_grad():
            o0.stage0_forward(feats[0][0], feats[0][1], params, STAGES[0][3])
        t1s0 = time.time() - t0
        t1sp = 0.0
        for s in (1, 2, 3):
            (L, R), (rm, tm) = feats[s], masks[s]
            frac = 6 if (s == 3 and budget_s < 6.0) else 1
            if frac > 1:
                rows = L.shape[2] // frac
                L, R, rm, tm = (t[..., :rows, :].contiguous() for t in (L, R, rm, tm))
            t0 = time.time()
            o, _, _ = oracle.spamat_forward(L, R, rm, tm, STAGES[s][3])
            oracle.spavar_forward(L, R, rm, tm, o, STAGES[s][3])
            t1sp += (time.time() - t0) * frac
        res["one_thread"] = {"value": 1.0 / (t1s0 + t1sp), "unit": "pairs/s", "cores": 1,
                             "sample": "1 pair: torch-CPU stage 0 %.1f s (whole) + C SpaMat+SpaVar stages 1-3 %.1f s (%s)"
                                       % (t1s0, t1sp, "whole" if budget_s >= 6.0 else
                                          "stage 3 timed on 90 of its 540 independent rows, x 6")}
    finally:
        oracle.set_num_threads(cores)
        torch.set_num_threads(cores)
    return res


def pack_mask_bits(mask):
    """float 0/1 [B,H,W] (device) -> int64 [B,H,ceil(W/64)], bit i of word w = pixel 64 w + i."""
    import torch
    B, H, W = mask.shape
    wpr = (W + 63) // 64
    z = torch.zeros(B, H, wpr * 64, dtype=torch.int64, device=mask.device)
    z[:, :, :W] = (mask != 0).long()
    sh = torch.arange(64, device=mask.device, dtype=torch.int64)
    return (z.view(B, H, wpr, 64) << sh).sum(-1).contiguous()


def live_traffic(config):
    """HBM-side bytes per launch of the dominant kernel, counted in THIS run: two child processes under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (counters need the profiler around the process; separate passes,
    no trace domains -- MI355X_MICROARCH.md) run three steps of this same script with every other leg switched off, and
    the per-launch means of the GEMM kernel are read back (tools/pmc_summary.py: FETCH_SIZE doubled on gfx950).  Returns
    (bytes, note) or (None, reason): any failure -- no rocprofv3, no counter access, a time-out -- leaves the committed
    profiles/traffic.json figure in place."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from pmc_summary import means
    except ImportError as e:
        return None, "tools/pmc_summary.py: %s" % e
    tmp = tempfile.mkdtemp(prefix="decnet_pmc_", dir="/tmp")
    res = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--config", str(config), "--steps", "3", "--warmup", "1",
                   "--no-cpu-baseline", "--no-e2e", "--no-train", "--no-density-sweep", "--no-alt", "--no-valu-floor",
                   "--no-live-traffic", "--no-blocks"]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=120, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s exited with %d" % (counter, r.returncode)
            m = {k: v for k, v in means(d, counter).items() if "wino_gemm_bf16x3" in k or "wino_gemm_persist<" in k or
                 "wino_gemm<" in k or "conv3d_k3_igemm" in k}
            if not m:
                return None, "no dominant-kernel rows in the %s pass" % counter
            tot = sum(n for _, n in m.values())
            res[counter] = sum(a * n for a, n in m.values()) / tot
        return 2.0 * 1024.0 * res["FETCH_SIZE"] + 1024.0 * res["WRITE_SIZE"], \
            "counted in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of 3 steps (FETCH_SIZE x 2 x 1024 + " \
            "WRITE_SIZE x 1024 bytes per launch, tools/pmc_summary.py)"
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError, IndexError) as e:
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cost_func_leg(feats, dev, iters=20):
    """The stage-0 branch (decnet_stage0_forward_cf) with each --cost_func of the reference (submodule.py:552-560) on the
    step's own stage-0 feature maps: "cor" is what demo.sh / eval.sh pass and what `value` is measured with; "ssd" changes
    the per-voxel arithmetic of the head kernel, "cat" adds conv_pre (two C x C products over the feature maps)."""
    import decnet_amd
    L, R = feats
    out = {}
    for cf in ("cor", "ssd", "cat"):
        st = decnet_amd.Stage0(make_regularizer(L.shape[1], dev, cost_func=cf))
        with torch.no_grad():
            out[cf + "_ms"] = round(time_kernel(lambda: st(L, R, STAGES[0][3]), iters, warm=5), 4)
    return out


def alt_gemm_leg():
    """The hot-path step once more in a child process with the fp32 MFMA Winograd GEMM (DECNET_WINO_GEMM=fp32; the
    switch is read once per process and changes the packed weights): value / ms_per_step / GEMM ms of round 2's
    default, beside this round's bf16x3 default."""
    import subprocess
    env = dict(os.environ, DECNET_WINO_GEMM="fp32")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "50", "--warmup", "8", "--no-cpu-baseline", "--no-train",
           "--no-density-sweep", "--no-e2e", "--no-alt", "--no-live-traffic", "--no-valu-floor", "--no-blocks"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                "wino_gemm_ms": d["roofline"]["ms"],
                "fp32_tflops": d["roofline"]["fp32_equivalent_tflops"],
                "fp32_mfma_frac": d["roofline"]["fp32_equivalent_tflops"] / MFMA_F32_PEAK_TF,
                "note": "DECNET_WINO_GEMM=fp32: the Winograd GEMMs on v_mfma_f32_16x16x4_f32 (wino_gemm_persist), the "
                        "default of rounds 1-2; fp32_mfma_frac = its flops / the 157.3 TFLOP/s fp32 MFMA peak"}
    except Exception as e:  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def e2e_model(dev):
    """The shipped network (demo.sh hyper-parameters), random init with seed 17; thold 0.5 so that the untrained mask
    generator produces mixed masks."""
    from decnet_amd.model import get_model
    torch.manual_seed(17)
    return get_model(name="sparsedensenetrefinementmask", max_disp=MAX_DISP, base_channels=8, cost_func="cor",
                     grad_method="detach", num_stage=4, down_scale=3, step=[-1., 1., 1., 1.],
                     samp_num=[-1., 12., 10., 6.], sample_spa_size_list=[-1, 3, 5, 7],
                     down_func_name="bicubic", weights=[1., 1., 1., 1.], if_overmask=False, skip_stage_id=4,
                     use_detail=True, thold=0.5).to(dev).eval()


def e2e_bench(B, dev, iters=5):
    """Whole network forward (random-init weights, demo.sh hyper-parameters, thold 0.5 so that the
    untrained mask generator produces mixed masks) on B synthetic 960x540 pairs padded to 972x540."""
    model = e2e_model(dev)
    g = torch.Generator(device=dev).manual_seed(17)
    left = torch.randn(B, 3, PAD_H, PAD_W, device=dev, generator=g)
    right = torch.randn(B, 3, PAD_H, PAD_W, device=dev, generator=g)
    with torch.no_grad():
        for _ in range(3):
            model(left, right)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            model(left, right)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        res = {"value": B / dt, "unit": "pairs/s", "ms_per_batch": 1e3 * dt, "batch": B,
               "note": "full graph: fused HIP small-channel 2-D convs + bf16x3 matrix-core many-channel convs + MIOpen for the stride-3 / small-image rest + the MI355X "
                       "hot-path kernels, eager launches"}
        # the same forward captured once into a HIP graph (static shapes) and replayed: no launch gaps
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(left, right)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = model(left, right)
            for _ in range(2):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                graph.replay()
            torch.cuda.synchronize()
            dtg = (time.perf_counter() - t0) / iters
            ref = model(left, right)
            ok = bool(torch.equal(out[0] if isinstance(out, (list, tuple)) else out,
                                  ref[0] if isinstance(ref, (list, tuple)) else ref))
            res["hip_graph"] = {"value": B / dtg, "ms_per_batch": 1e3 * dtg, "replay_equals_eager": ok}
        except Exception as e:                          # capture is an optimisation, never a requirement
            res["hip_graph"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        # algorithmic work of one forward, per kernel family: every Conv2dUnit / Deconv2dUnit launch reports its own
        # (decnet_amd.model.TALLY); stage 0 and the SpaMat / SpaVar passes in closed form (e2e_kernel_table)
        from decnet_amd import model as M
        M.TALLY = []
        try:
            model(left, right)
            torch.cuda.synchronize()
            agg = {}
            for t in M.TALLY:
                a = agg.setdefault(t["family"], {"launches": 0, "flops": 0.0, "bytes": 0.0})
                a["launches"] += 1
                a["flops"] += t["flops"]
                a["bytes"] += t["bytes"]
            res["unit_work"] = agg
        finally:
            M.TALLY = None
    return res


# rocprofv3 kernel name -> kernel family of the end-to-end table, first match wins
E2E_FAMILIES = (
    ("wino_gemm", "stage 0: Winograd GEMMs (7 x Conv3d 216->216) + the ASPP tap GEMM [bf16x3 on the bf16 matrix pipe]"),
    ("wino_mid_transform", "stage 0: fused output/input Winograd transforms between the layers"),
    ("wino_head_transform", "stage 0: cost volume formed on chip + first input transform"),
    ("wino_", "stage 0: other Winograd transforms"),
    ("cout1_", "stage 0: Conv3d 216->1 + soft-argmax"),
    ("spamat_fwd", "cost-volume pass: fused SpaMat + SpaVar, stages 1-3"),
    ("conv2d_mfma", "2-D trunk: many-channel 3x3 / 1x1 convolutions [bf16x3 on the bf16 matrix pipe]"),
    ("conv2d_f32m", "2-D trunk: few-channel 3x3 convolutions [fp32, v_mfma_f32_4x4x1]"),
    ("conv2d_small", "2-D trunk: few-channel convolutions [fp32 FMA]"),
    ("deconv2d_k3s3", "2-D trunk: stride-3 few-channel transposed convolutions"),     # before its substring "conv2d_k3s3"
    ("conv2d_k3s3", "2-D trunk: stride-3 few-channel convolutions"),
    ("detail_mask", "mask generator tail (sigmoid > thold, bit-packed masks)"),
    ("warp_disparity", "Refinement: warp of the right features"),
    ("miopen", "library convolutions (MIOpen / Tensile)"), ("Cijk", "library convolutions (MIOpen / Tensile)"),
    ("igemm", "library convolutions (MIOpen / Tensile)"), ("MIOpen", "library convolutions (MIOpen / Tensile)"),
)
# Unit kinds (decnet_amd.model._tally) -> the family prefix above
UNIT_KIND_PREFIX = {"mfma": "conv2d_mfma", "mfma_s3": "conv2d_mfma", "mfma_deconv": "conv2d_mfma", "conv": "conv2d_small",
                    "conv_s3": "conv2d_k3s3", "deconv": "deconv2d_k3s3", "library": "miopen"}


def e2e_kernel_table(B, unit_work, top=5):
    """The five largest kernel families of ONE end-to-end forward with their roofline position: a child process under
    `rocprofv3 --kernel-trace` (the program directly behind `--`) runs tools/e2e_profile.py, the launches between the
    last two stage-0 head kernels are one steady-state forward; algorithmic flops / bytes per family from the layers'
    own report (unit_work) or, for stage 0 and the cost-volume pass, in closed form.  frac = the larger
    of (algorithmic bytes / time) / 8 TB/s and (executed flops / time) / the pipe's dense peak."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return {"error": "rocprofv3 not on PATH"}
    tmp = tempfile.mkdtemp(prefix="decnet_e2e_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "e2e", "--",
                            sys.executable, os.path.join(ROOT, "tools", "e2e_profile.py"), str(B)], cwd="/tmp", env=env,
                           timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = glob.glob(tmp + "/**/*kernel_trace.csv", recursive=True)
        if r.returncode != 0 or not files:
            return {"error": "rocprofv3 --kernel-trace exited with %d" % r.returncode}
        rows = sorted(csv.DictReader(open(files[0])), key=lambda x: int(x["Start_Timestamp"]))
        idx = [i for i, x in enumerate(rows) if "wino_head_transform" in x["Kernel_Name"]] or \
              [i for i, x in enumerate(rows) if "costvol_cor_ndhwc" in x["Kernel_Name"]]
        if len(idx) < 2:
            return {"error": "no two stage-0 head launches in the trace"}
        # one forward = the launches between two consecutive stage-0 head kernels; of the last few, the one with the
        # shortest span (a HIP-graph replay: no host gaps, no synchronisation in between)
        cands = [(int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"]), a, b)
                 for a, b in zip(idx[-7:-1], idx[-6:]) if b - a > 50]
        if not cands:
            return {"error": "no complete forward between two stage-0 head launches"}
        _, a, b = min(cands)
        seg = rows[a:b]
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fam = collections.OrderedDict()
    for x in seg:
        name = x["Kernel_Name"]
        key = next((p for p, _ in E2E_FAMILIES if p in name), None)
        label = dict(E2E_FAMILIES).get(key, name[:70])
        f = fam.setdefault(key or name[:70], {"kernel": label, "calls": 0, "ms": 0.0, "names": set()})
        f["calls"] += 1
        f["ms"] += (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6
        f["names"].add(name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48])
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    busy = sum(f["ms"] for f in fam.values())
    # closed-form algorithmic work of the non-Unit families (both views are one batch of 2 B in the extractor)
    C0, H0, W0, D0 = STAGES[0]
    nt = B * ((D0 + 3) // 4) * ((H0 + 3) // 4) * ((W0 + 3) // 4)
    cp = (C0 + 15) // 16 * 16
    aspp_flops = 2.0 * (2 * B * H0 * W0) * C0 * C0 * 28          # 1x1 + three dilated 3x3 branches = 28 taps
    work = {
        "wino_gemm": {"flops": 7 * 2.0 * 216 * nt * C0 * C0 + aspp_flops, "executed_x": 6.0, "peak_tf": MFMA_BF16_PEAK_TF,
                      "bytes": 7 * (2.0 * 216 * nt * cp * 4 + 216 * cp * 224 * 6) + aspp_flops / (2.0 * C0) * 4 * 2},
        "wino_mid_transform": {"bytes": 6 * 2.0 * 216 * nt * cp * 4},
        "wino_head_transform": {"bytes": 216.0 * nt * cp * 4 + 2.0 * B * C0 * H0 * W0 * 4},
        "spamat_fwd": {"bytes": sum(4.0 * B * H * W * (2 * C + 2 + 4) for (C, H, W, D) in STAGES[1:])},
    }
    for kind, w in (unit_work or {}).items():
        pre = UNIT_KIND_PREFIX.get(kind)
        if pre:
            t = work.setdefault(pre, {"flops": 0.0, "bytes": 0.0})
            t["flops"] = t.get("flops", 0.0) + w["flops"]
            t["bytes"] = t.get("bytes", 0.0) + w["bytes"]
    for pre in ("conv2d_mfma",):
        if pre in work:
            work[pre].update(executed_x=6.0, peak_tf=MFMA_BF16_PEAK_TF)      # three bf16 terms per operand, six products
    # the few-channel kernels share their layers' report: conv2d_f32m and conv2d_small are two kernels of the "conv" kind
    if "conv2d_small" in work and "conv2d_f32m" in fam:
        both = fam["conv2d_f32m"]["ms"] + fam.get("conv2d_small", {"ms": 0.0})["ms"]
        share = fam["conv2d_f32m"]["ms"] / both if both else 0.0
        w = work["conv2d_small"]
        work["conv2d_f32m"] = {"flops": w["flops"] * share, "bytes": w["bytes"] * share, "peak_tf": MFMA_F32_PEAK_TF,
                               "split_by_time": True}
        work["conv2d_small"] = {"flops": w["flops"] * (1 - share), "bytes": w["bytes"] * (1 - share),
                                "peak_tf": MFMA_F32_PEAK_TF, "split_by_time": True}
    table = []
    for key, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:top]:
        row = {"kernel": f["kernel"], "kernel_names": sorted(f["names"])[:4], "calls": f["calls"], "ms": f["ms"],
               "share_of_busy": f["ms"] / busy}
        w = work.get(key)
        if w:
            fr = []
            if w.get("bytes"):
                row["algorithmic_bytes"] = w["bytes"]
                row["hbm_frac"] = w["bytes"] / f["ms"] / 1e6 / HBM_PEAK_GBS
                fr.append(("hbm", row["hbm_frac"]))
            if w.get("flops"):
                ex = w["flops"] * w.get("executed_x", 1.0)
                peak = w.get("peak_tf", MFMA_F32_PEAK_TF)
                row["algorithmic_flops"] = w["flops"]
                row["executed_flops"] = ex
                row["matrix_or_fp32_peak_tflops"] = peak
                row["compute_frac"] = ex / f["ms"] / 1e9 / peak
                fr.append(("compute", row["compute_frac"]))
            if w.get("split_by_time"):
                row["note"] = "conv2d_small and conv2d_f32m run layers of one kind: their joint algorithmic work is split by time"
            if fr:
                row["bound"], row["frac"] = max(fr, key=lambda t: t[1])
        table.append(row)
    return {"one_forward": {"kernels": len(seg), "span_ms": span, "busy_ms": busy}, "top_kernels": table,
            "definition": "one steady-state forward out of a rocprofv3 --kernel-trace child run; frac = max(algorithmic "
                          "bytes / time / 8 TB/s, executed flops / time / dense peak of the pipe the family runs on)"}


def e2e_acc2_leg():
    """The end-to-end forward once more in a child process with DECNET_CONV2D_ACC=2 (the bf16x3 trunk kernels with a
    second accumulator set: error below an fp32 fma chain's; the switch is read once per process): what reference-grade
    fp32 in the many-channel layers costs."""
    import subprocess
    env = dict(os.environ, DECNET_CONV2D_ACC="2")
    code = ("import sys, json, torch; sys.path.insert(0, %r); import bench; "
            "r = bench.e2e_bench(%d, torch.device('cuda:0')); r.pop('unit_work', None); print(json.dumps(r))" % (ROOT, DEFAULT_B))
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        hg = d.get("hip_graph", {})
        return {"value": max(d["value"], hg.get("value", 0.0)), "unit": "pairs/s",
                "ms_per_batch": min(d["ms_per_batch"], hg.get("ms_per_batch", 1e9)),
                "note": "DECNET_CONV2D_ACC=2: conv2d_mfma with two accumulator sets (csrc/conv2d_mfma_acc2.hip); accuracy "
                        "gate: tests/test_inputdata_gpu.py::test_two_accumulator_trunk_is_as_close_to_float64_as_the_reference"}
    except Exception as e:  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def e2e_cpu_baseline(budget_s=30.0):
    """SURVEY 8d metric (1) on the host: the SAME graph -- decnet_amd.model's modules on CPU tensors (their torch
    fallbacks: the reference's own Conv2d / BatchNorm / grid_sample / interpolate calls), stage 0 through
    oracle/stage0.py and SpaMat / SpaVar through the C + OpenMP oracle -- on one synthetic 972x540 pair, all host cores.
    A reported baseline, like `cpu_baseline`; this leg and no product path touches oracle/."""
    import oracle
    from oracle import stage0 as o0
    from decnet_amd import model as M
    oracle.build()
    cores = oracle.num_threads()
    torch.set_num_threads(cores)
    model = e2e_model(torch.device("cpu"))
    params = o0.params_from_module(model.cost_regularizer)

    def spamatvar_cpu(L, R, lm, rm, D, out=None):
        o, s, mx = oracle.spamat_forward(L, R, lm, rm, D)
        v, _, _ = oracle.spavar_forward(L, R, lm, rm, o, D)
        return tuple(torch.from_numpy(a) for a in (o, v, s, mx))

    def stage0_cpu(left, right, max_disp, return_reg=False):
        pred, reg, _ = o0.stage0_forward(left, right, params, max_disp)
        return (pred, reg) if return_reg else pred

    saved = M.spamatvar_forward
    M.spamatvar_forward = spamatvar_cpu
    model.cost_regularizer.stage0 = stage0_cpu
    try:
        g = torch.Generator().manual_seed(17)
        left = torch.randn(1, 3, PAD_H, PAD_W, generator=g)
        right = torch.randn(1, 3, PAD_H, PAD_W, generator=g)
        with torch.no_grad():
            t0 = time.time()
            model(left, right)                           # warm-up: thread pools, library load
            warm = time.time() - t0
            n, t = 0, 0.0
            while n < 4 and (n == 0 or warm + t + t / n < budget_s):
                t0 = time.time()
                model(left, right)
                t += time.time() - t0
                n += 1
    finally:
        M.spamatvar_forward = saved
    return {"value": n / t, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d pair(s) %dx%d max_disp %d after 1 warm-up pair (%.1f s): whole graph on torch-CPU + the "
                      "C/OpenMP SpaMat/SpaVar oracle + oracle/stage0.py, %.2f s per pair" % (n, PAD_W, PAD_H, MAX_DISP, warm, t / n)}


def train_leg(dev, B=4, iters=30):
    """BASELINE config 5 (per-GPU share: 4 pairs of 972x540): SpaMat forward + backward at stages 1-3, mask
    densities 1.0, 0.5 and 0.1.  Kernel times: the C-ABI entry points on preallocated buffers (events over
    back-to-back launches); step time: the same through SpaMatFunction.apply / .backward (allocations and
    autograd included, no host sync inside the loop).  Backward bytes: 4*B*H*W*(4C + 6)."""
    import decnet_amd
    from decnet_amd import ops
    mod = decnet_amd.SpaMat()
    res = []
    for dens in (1.0, 0.5, 0.1):
        feats, masks = make_inputs(B, dev, dens, seed=555)
        row = {"mask_density": dens, "stages": []}
        step_ms, kernel_ms, graphed = 0.0, 0.0, []
        for s in (1, 2, 3):
            C, H, W, D = STAGES[s]
            L, R = feats[s]
            rm, tm = masks[s]
            go = torch.randn(B, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
            o, ss, mc = (torch.empty(B, H, W, device=dev) for _ in range(3))
            gl, gr = torch.empty_like(L), torch.empty_like(R)
            with torch.no_grad():
                tf = time_kernel(lambda: ops.spamat_forward(L, R, rm, tm, o, ss, mc, D), iters)
                tb = time_kernel(lambda: ops.spamat_backward(L, R, rm, tm, o, ss, mc, go, gl, gr, D), iters)
            Lg, Rg = L.clone().requires_grad_(), R.clone().requires_grad_()

            def step():
                Lg.grad = Rg.grad = None
                mod(Lg, Rg, rm, tm, D).backward(go)
            ts = time_kernel(step, iters)
            nb = 4.0 * B * H * W * (4 * C + 6)
            row["stages"].append({"stage": s, "fwd_kernel_ms": tf, "bwd_kernel_ms": tb, "autograd_step_ms": ts,
                                  "bwd_GBps": nb / tb / 1e6, "bwd_frac_hbm": nb / tb / 1e6 / HBM_PEAK_GBS})
            step_ms += ts
            kernel_ms += tf + tb
            graphed.append((Lg, Rg, rm, tm, D, go))
        row["fwd_bwd_eager_ms"] = step_ms
        row["fwd_bwd_kernels_ms"] = kernel_ms
        if dens == 1.0:
            # the dense stage-3 backward against ITS bound: instruction issue of its band tiles (live microbenchmark)
            C, H, W, D = STAGES[3]
            xt, nt = (W + 15) // 16, (D - 1 + 15) // 16 + 1
            tiles = B * H * sum(min(x, nt - 1) + 1 for x in range(xt))
            vf = bwd_valu_floor(tiles)
            tb3 = row["stages"][-1]["bwd_kernel_ms"]
            if vf:
                row["roofline_valu"] = {"bound": "valu", "achieved": vf["floor_ms"], "peak": tb3,
                                        "unit": "ms (floor / measured)", "frac": vf["floor_ms"] / tb3, "ms": tb3,
                                        "kernel": "spamat backward, stage 3, dense rows (spamat_bwd_rowb + marker launches)",
                                        "detail": vf}
        # the whole step (SpaMatFunction forward + backward, stages 1-3) as ONE HIP-graph replay: the same
        # autograd.Function, captured once (decnet_amd.graphs.GraphedStep) -- no Python / allocator work per step
        try:
            from decnet_amd.graphs import GraphedStep

            def whole():
                for (a, b2, c, d2, dd, g) in graphed:
                    mod(a, b2, c, d2, dd).backward(g)
            gs = GraphedStep(whole, grads_of=[t for item in graphed for t in item[:2]])
            row["fwd_bwd_ms"] = time_kernel(gs, iters)
            row["fwd_bwd_ms_is"] = "one HIP-graph replay of the three SpaMatFunction forward + backward calls"
        except Exception as e:  # noqa: BLE001 -- the eager figure stands in, and says so
            row["fwd_bwd_ms"] = step_ms
            row["fwd_bwd_ms_is"] = "eager (graph capture failed: %s: %s)" % (type(e).__name__, str(e)[:160])
        step_ms = row["fwd_bwd_ms"]
        row["pairs_per_s"] = B / step_ms * 1e3
        res.append(row)
    out = {"workload": "BASELINE config 5 per-GPU share: SpaMat forward+backward, stages 1-3, "
                       "%d pairs 972x540 max_disp 216" % B,
           "by_density": res}
    if "roofline_valu" in res[0]:
        out["roofline_valu"] = res[0].pop("roofline_valu")
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) under
    torch.distributed.run and pass their output through; rank 0 prints the JSON line.  This parent has not
    touched the GPU (no HIP call before this point) and it does not exec: the ranks are child processes."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def selftest_cpu(args, world, rank):
    """--selftest-cpu: the N-rank plumbing (rendezvous, pair sharding, the disparity all-gather, the bucketed
    gradient all-reduce) on the gloo backend without a GPU.  No kernel runs and nothing is measured: `value`
    is null.  Used by tests/test_bench_cpu.py to cover the self-launch path in the build container."""
    from decnet_amd import dist as dd
    torch.distributed.init_process_group("gloo")
    assert torch.distributed.get_world_size() == args.gpus, "rendezvous has %d ranks, --gpus %d" % (
        torch.distributed.get_world_size(), args.gpus)
    B = args.pairs_per_gpu or 2
    s, e = dd.shard_range(world * B, rank, world)
    local = torch.arange(s, e, dtype=torch.float32).view(-1, 1, 1).expand(-1, 4, 6).contiguous()
    got = dd.gather_disparity(local, n_pairs=world * B)
    ok = bool(torch.equal(got[:, 0, 0], torch.arange(world * B, dtype=torch.float32)))
    gb = dd.GradBuckets(4096, n_buckets=4)
    for i in range(len(gb)):
        gb.bucket(i).fill_(float(rank))
        gb.reduce_async(i)
    gb.wait()
    ok = ok and bool(torch.allclose(gb.flat, torch.full_like(gb.flat, (world - 1) / 2.0)))
    t = torch.tensor([1.0 if ok else 0.0])
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"metric": "selftest (no GPU work)", "value": None, "unit": "pairs/s", "n_gpus": world,
                          "steps": 0, "warmup": 0, "selftest": True, "ok": bool(t.item() == 1.0),
                          "backend": "gloo"}), flush=True)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    return 0 if t.item() == 1.0 else 1


class TrainShare:
    """BASELINE config 5, the hot path's share of one data-parallel training step on this rank's pairs:
    SpaMat forward (sparse_matching_forward) and backward (ref + tar gradients) at stages 1-3 through the C ABI
    on preallocated buffers, and the all-reduce of the network's 52.7 MB of parameter gradients in 4 buckets,
    each launched right after one of the backward stages so that RCCL moves it while the next stage's kernels
    run (the gradients in the buffer are synthetic: the 2-D trunk's backward is PyTorch's, outside this path)."""

    def __init__(self, B, dev, density, world):
        from decnet_amd import dist as dd, ops
        self.ops, self.B, self.world = ops, B, world
        self.coll = world > 1 or dd.force_collective()
        self.feats, self.masks = make_inputs(B, dev, density, seed=555 + 1000 * int(os.environ.get("RANK", 0)))
        self.buf = {}
        for s in (1, 2, 3):
            C, H, W, D = STAGES[s]
            g = torch.Generator(device=dev).manual_seed(9 + s)
            self.buf[s] = dict(go=torch.randn(B, H, W, device=dev, generator=g),
                               o=torch.empty(B, H, W, device=dev), ss=torch.empty(B, H, W, device=dev),
                               mc=torch.empty(B, H, W, device=dev), gl=torch.empty(B, C, H, W, device=dev),
                               gr=torch.empty(B, C, H, W, device=dev))
        self.grads = dd.GradBuckets(N_PARAMS, n_buckets=4, device=dev)
        self.grads.flat.normal_(generator=torch.Generator(device=dev).manual_seed(3))

    def step(self):
        ops = self.ops
        for s in (1, 2, 3):
            (L, R), (rm, tm), b = self.feats[s], self.masks[s], self.buf[s]
            ops.spamat_forward(L, R, rm, tm, b["o"], b["ss"], b["mc"], STAGES[s][3])
        nb = len(self.grads)
        for k, s in enumerate((3, 2, 1)):             # backward runs fine-to-coarse
            (L, R), (rm, tm), b = self.feats[s], self.masks[s], self.buf[s]
            ops.spamat_backward(L, R, rm, tm, b["o"], b["ss"], b["mc"], b["go"], b["gl"], b["gr"], STAGES[s][3])
            if self.coll:
                for i in ([0], [1], list(range(2, nb)))[k]:
                    self.grads.reduce_async(i)
        if self.coll:
            self.grads.wait()

    def drain(self):
        pass


def timed_region(step, drain, warmup, steps, world, dev):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
    def barrier():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    if torch.distributed.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def value_blocks(step, drain, steps, world, dev, units_per_step, nblocks=5):
    """The timed region again, `nblocks` times back to back in the same process (each block = exactly `steps` steps
    between barrier + synchronize, like `value`'s own): min / median / max of the per-block rate, so that a
    box-to-box or clock difference can be told from a code difference on one run."""
    vals = []
    for _ in range(nblocks):
        el = timed_region(step, drain, 0, steps, world, dev)
        vals.append(units_per_step * steps / el)
    sv = sorted(vals)
    return {"n_blocks": nblocks, "steps_per_block": steps, "min": sv[0], "median": sv[len(sv) // 2], "max": sv[-1],
            "blocks": vals, "unit": "pairs/s",
            "note": "the same timed region repeated after `value`'s own; `value` is the FIRST block after the warmup"}


def smi_under_load(step, drain, min_s=1.5):
    """Shader clock and package power WHILE the hot path runs: a rocm-smi child is started, the step loop keeps the GPU
    busy until the child has exited (rank 0, one GPU; never inside a timed region).  None if rocm-smi is not there."""
    import shutil
    import subprocess
    import threading
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    box = {}

    def run():
        try:
            time.sleep(0.4)                              # let the loop below reach its steady state first
            r = subprocess.run([smi, "-d", str(torch.cuda.current_device()), "--showclocks", "--showpower", "--json"],
                               capture_output=True, text=True, timeout=30)
            box["raw"] = r.stdout
        except Exception as e:                           # noqa: BLE001 -- an extra key, never fatal
            box["err"] = "%s: %s" % (type(e).__name__, str(e)[:120])
    th = threading.Thread(target=run)
    t0 = time.perf_counter()
    th.start()
    n = 0
    while th.is_alive() or time.perf_counter() - t0 < min_s:
        for _ in range(20):
            step()
        drain()
        torch.cuda.synchronize()
        n += 20
    th.join()
    out = {"steps_run_meanwhile": n, "busy_s": time.perf_counter() - t0}
    try:
        card = next(iter(json.loads(box["raw"]).values()))
        # rocm-smi's own keys, e.g. "sclk clock speed:" -> "(2100Mhz)", "sclk clock level:" -> "1",
        # "Current Socket Graphics Package Power (W)" -> "1297.0": kept verbatim
        out["rocm_smi"] = {k: v for k, v in card.items()
                           if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "socclk", "power"))}
    except Exception as e:                               # noqa: BLE001
        out["error"] = box.get("err") or "%s: %s" % (type(e).__name__, str(e)[:120])
        out["raw"] = (box.get("raw") or "")[:300]
    return out


def main_train(args, B, dev, world, rank):
    """--config 5 (see TrainShare)."""
    ts = TrainShare(B, dev, args.mask_density, world)
    with torch.no_grad():
        elapsed = timed_region(ts.step, ts.drain, args.warmup, args.steps, world, dev)
        t_ar = None
        if ts.coll:                                      # the gradient all-reduce alone, not overlapped: a
            def ar():                                    # collective, so EVERY rank runs it (outside the timed region)
                for i in range(len(ts.grads)):
                    ts.grads.reduce_async(i)
                ts.grads.wait()
            t_ar = time_kernel(ar, 10, warm=3)
        if rank == 0:
            ms_step = 1e3 * elapsed / args.steps
            C, H, W, D = STAGES[3]
            (L, R), (rm, tm), b = ts.feats[3], ts.masks[3], ts.buf[3]
            tb = time_kernel(lambda: ts.ops.spamat_backward(L, R, rm, tm, b["o"], b["ss"], b["mc"], b["go"], b["gl"],
                                                            b["gr"], D), 30)
            tf = time_kernel(lambda: ts.ops.spamat_forward(L, R, rm, tm, b["o"], b["ss"], b["mc"], D), 30)
            nb = 4.0 * B * H * W * (4 * C + 6)
            traffic = {}
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                    traffic = json.load(f)
            except (OSError, ValueError):
                pass
            out = {
                "metric": "stereo pairs/sec, BASELINE config 5 share (SpaMat forward+backward, stages 1-3, + "
                          "all-reduce of a SYNTHETIC 52.7 MB gradient buffer: the 2-D trunk's backward is outside the path)",
                "value": world * B * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "BASELINE %s: %d pairs per GPU, feature maps of the 4-stage/scale-3 net; the "
                                       "gradient buffer is synthetic (the 2-D trunk's backward is outside the path)"
                                       % (CONFIG_NAME, B),
                           "pairs_per_gpu": B, "mask_density": args.mask_density, "grad_bytes": 4 * N_PARAMS,
                           "grad_buckets": len(ts.grads),
                           "parallelism": "dp%d (pairs sharded, bucketed all_reduce of gradients)" % world},
                "roofline": {"bound": "hbm", "achieved": nb / tb / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": nb / tb / 1e6 / HBM_PEAK_GBS,
                             "traffic": traffic.get("spamat_bwd_stage3", {}).get("total_bytes"),
                             "kernel": "spamat backward (ref + tar gradient launches), stage 3", "ms": tb,
                             "bytes_per_launch": nb, "fwd_ms": tf},
                "allreduce_alone_ms": t_ar,
            }
            if ts.coll:
                out["collective"] = {"backend": torch.distributed.get_backend(), "world": world,
                                     "forced_at_world_1": world == 1, "allreduce_4_buckets_alone_ms": t_ar}
            print(json.dumps(out), file=_JSON_OUT or sys.stdout, flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json config (2 = the metric's; 3, 4: the other single-GPU-shard shapes)")
    ap.add_argument("--pairs-per-gpu", type=int, default=0, help="default: the config's (8 / 4 / 1 / 4)")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="no GPU: exercise the N-rank launch, sharding and collectives on gloo (value null)")
    ap.add_argument("--mask-density", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0,
                    help="seconds of host work the cpu_baseline leg may take (its sample says what it covered)")
    ap.add_argument("--no-density-sweep", action="store_true",
                    help="skip the extra cost-volume timings at mask densities 0.3 .. 0.02 (PMC passes: keeps "
                         "every launch of a kernel the same work)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run SpaMat/SpaVar after stage 0 on the same stream instead of beside it on a second one "
                         "(profiling runs: a kernel trace then shows every kernel alone on the GPU, the way the "
                         "roofline leg times it)")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the extra 'train' object (config 5: SpaMat forward+backward, stages 1-3)")
    ap.add_argument("--e2e", action="store_true", help="(default at 1 GPU) see --no-e2e")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the 'alt_wino_gemm_fp32' object: the same hot-path step in a child process with "
                         "DECNET_WINO_GEMM=fp32 (the Winograd GEMMs on fp32 MFMA, round 2's default)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="take roofline.traffic from profiles/traffic.json instead of counting it in two rocprofv3 --pmc "
                         "child passes of this run (profiling runs; the child passes themselves)")
    ap.add_argument("--no-valu-floor", action="store_true",
                    help="skip the live VALU-floor microbenchmark (a child process; profiling runs)")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise RCCL (backend nccl) and run the per-step collectives (all-gather of the disparity "
                         "maps; config 5: the bucketed gradient all-reduce) even with ONE rank: the N > 1 code path on a "
                         "one-GPU box.  Adds a 'collective' object with the collective's latency and an equality check")
    ap.add_argument("--no-e2e-table", action="store_true",
                    help="skip e2e.roofline: the per-kernel-family table of one forward (a child process under "
                         "rocprofv3 --kernel-trace)")
    ap.add_argument("--precondition-steps", type=int, default=200,
                    help="untimed hot-path steps BEFORE the W warm-up steps (0: none; 200 = about 0.3 s): brings a fresh "
                         "process to the clocks and caches of a running one; reported as preconditioning_steps")
    ap.add_argument("--no-blocks", action="store_true",
                    help="skip value_blocks (the timed region repeated 5 more times: min / median / max) and the "
                         "rocm-smi sample of shader clock and package power under load")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the extra 'e2e' object: the whole inference graph (decnet_amd.model: the 2-D "
                         "trunk around the hot path) timed on the same batch, eager and as a HIP-graph replay")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: one process per GPU, started here, before anything touches HIP
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.selftest_cpu:
        raise SystemExit(selftest_cpu(args, world, rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.force_collective:
        os.environ["DECNET_FORCE_COLLECTIVE"] = "1"
    if world > 1 or args.force_collective:
        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the ONE JSON
        # line (everything else this process or its libraries print goes to stderr)
        global _JSON_OUT
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:             # plain `python bench.py --force-collective`: a world of one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.distributed.init_process_group("nccl", device_id=dev)
        if torch.distributed.get_world_size() != args.gpus:
            raise SystemExit("RCCL sees %d ranks, --gpus %d" % (torch.distributed.get_world_size(), args.gpus))

    set_config(args.config)
    B = args.pairs_per_gpu or DEFAULT_B
    if args.config == 5:
        return main_train(args, B, dev, world, rank)
    hp = HotPath(B, dev, args.mask_density, world)
    hp.overlap = not args.no_overlap

    def barrier():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # nothing but the work inside the timed region (no events); every all-gather has landed when it ends
        # Steady state before the contract's W warm-up steps: a process that has just created its context runs its first
        # ~100 ms 3 % slower (clock ramp, first touches; value_blocks of round 6: 4 939 in the first block of 20 steps
        # after 5 warm-up steps, 5 070 - 5 100 in the five blocks behind it).  A fixed number of untimed steps, stated in
        # the line; the W warm-up steps and the K timed steps follow unchanged.
        for _ in range(args.precondition_steps):          # a COUNT, not a time: every rank runs the same collectives
            hp.step()
        hp.drain()
        torch.cuda.synchronize()
        elapsed = timed_region(hp.step, hp.drain, args.warmup, args.steps, world, dev)
        blocks = None if args.no_blocks else value_blocks(hp.step, hp.drain, args.steps, world, dev, world * B)
        smi = smi_under_load(hp.step, hp.drain) if (rank == 0 and world == 1 and not args.no_blocks) else None
        # per-stage breakdown from a few extra steps with events around stage 0 and stage 3 (an event
        # record costs ~20 us of pipeline bubble each, so they stay out of the timed region)
        nev = min(args.steps, 10)
        ev = [{k: torch.cuda.Event(enable_timing=True) for k in ("s0_beg", "s0_end", "s3_beg", "s3_end")}
              for _ in range(nev)]
        for i in range(nev):
            hp.step(ev[i])
        hp.drain()
        barrier()
    if rank == 0:
        # HBM-side traffic per launch comes from a separate rocprofv3 --pmc pass (counters cannot be
        # read from inside the process); profiles/traffic.json records the command and corrections.
        traffic = {}
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic = json.load(f)
        except (OSError, ValueError):
            pass
        ms_step = 1e3 * elapsed / args.steps
        pairs = world * B * args.steps
        # --- dominant kernel, timed live with events on the launch stream -------------------
        C0, H0, W0, D0 = STAGES[0]
        M = B * D0 * H0 * W0
        conv_flop = 2.0 * 27 * C0 * C0 * M                      # direct-convolution flops of one layer
        s0_ms = sum(e["s0_beg"].elapsed_time(e["s0_end"]) for e in ev) / len(ev)
        with torch.no_grad():
            cv, a, _ = hp.reg.stage0_buffers(dev, B, C0, H0, W0, D0)     # left by the steps above
            P = hp.reg.prepare(D0)
            from decnet_amd import _lib
            from decnet_amd.stage0 import conv_algo, WINO_VARIANT
            L = _lib.lib()
            st = torch.cuda.current_stream().cuda_stream
            p0 = P[0]
            gemm_mult, gemm_peak, gemm_arith = 1.0, MFMA_F32_PEAK_TF, "fp32 MFMA (v_mfma_f32_16x16x4_f32)"
            if conv_algo(D0) in WINO_VARIANT:
                # one Conv3d layer = input transform + batched GEMMs (one per transform point) + output
                # transform; the GEMM kernel (wino_gemm) is the dominant kernel of the step
                var = WINO_VARIANT[conv_algo(D0)]
                od, oh, npts = ((2, 2, 64), (2, 4, 144), (4, 4, 216))[var]
                nt = B * ((D0 + od - 1) // od) * ((H0 + oh - 1) // oh) * ((W0 + oh - 1) // oh)
                wsp = torch.empty(L.decnet_conv3d_wino_workspace_floats(B, D0, H0, W0, C0, C0, var), device=dev)
                cp = (C0 + 15) // 16 * 16
                V, Mw = wsp[:npts * nt * cp], wsp[npts * nt * cp:]
                # (the step forms the cost volume on chip where the fused stack covers the shape: write it out for
                # the per-layer timings below)
                Lf0, Rf0 = hp.feats[0]
                _lib.check(L.decnet_costvol_forward(Lf0.data_ptr(), Rf0.data_ptr(), cv.data_ptr(), B, C0, H0, W0, D0, st),
                           "decnet_costvol_forward")
                layer_unfused_ms = layer_ms = time_kernel(lambda: L.decnet_conv3d_wino_bn_act(
                    cv.data_ptr(), p0["u"].data_ptr(), p0["scale"].data_ptr(), p0["shift"].data_ptr(), None,
                    a.data_ptr(), wsp.data_ptr(), B, D0, H0, W0, C0, C0, 1, var, st), 30)
                # the seven C -> C layers as the step runs them: one fused stack (output transform of layer i and input
                # transform of layer i + 1 in one kernel, activations between the layers in LDS)
                stack_ms = None
                nst = L.decnet_conv3d_wino_stack_workspace_floats(B, D0, H0, W0, C0, var)
                if nst and os.environ.get("DECNET_WINO_STACK", "1") != "0":
                    import ctypes
                    wst = torch.empty(nst, device=dev)
                    arr = ctypes.c_void_p * 7
                    us, scs, shs = (arr(*[P[i][k].data_ptr() for i in range(7)]) for k in ("u", "scale", "shift"))
                    stack_ms = time_kernel(lambda: L.decnet_conv3d_wino_stack_bn_act(
                        cv.data_ptr(), us, scs, shs, 7, 1, 4, a.data_ptr(), wst.data_ptr(), B, D0, H0, W0, C0, var, st), 20)
                    layer_ms = stack_ms / 7.0
                conv_ms = time_kernel(lambda: L.decnet_conv3d_wino_gemm(
                    V.data_ptr(), p0["u"].data_ptr(), Mw.data_ptr(), nt, C0, C0, var, st), 60)
                kern_flop = 2.0 * npts * nt * C0 * C0
                bf16x3 = C0 == 216 and os.environ.get("DECNET_WINO_GEMM", "") in ("", "bf16x3")
                if bf16x3:
                    gemm_mult, gemm_peak = 6.0, MFMA_BF16_PEAK_TF
                    gemm_arith = "fp32 operands as three bf16 terms (round to nearest), six products on " \
                                 "v_mfma_f32_16x16x32_bf16, fp32 accumulation"
                kern_name = "%s (%d x [%d x %d] x [%d x %d], %s Conv3d 216->216)" % (
                    "wino_gemm_bf16x3" if bf16x3 else "wino_gemm", npts, nt, C0, C0, C0,
                    ("Winograd F(2,3)^3", "Winograd F(2,3)xF(4,3)^2", "Winograd F(4,3)^3")[var])
                tkey = "wino_gemm"
                # the launch's algorithmic HBM bytes: V in, M out (216 points x tiles x 224-padded channels x 4 B each)
                # and the weights it reads (U^T as three bf16 terms per value for the bf16x3 kernel, else fp32)
                kern_bytes = 2.0 * npts * nt * cp * 4 + npts * cp * 224 * (6 if bf16x3 else 4)
            else:
                kern_bytes = None
                stack_ms = layer_unfused_ms = None
                conv_ms = layer_ms = time_kernel(lambda: L.decnet_conv3d_bn_act(
                    cv.data_ptr(), p0["w"].data_ptr(), p0["scale"].data_ptr(), p0["shift"].data_ptr(), None,
                    a.data_ptr(), B, D0, H0, W0, C0, C0, 1, st), 10)
                kern_flop = conv_flop
                kern_name = "conv3d_k3_igemm (216->216, 3^3, M=%d)" % M
                tkey = "conv3d_k3_igemm"
            # --- cost-volume pass (fused SpaMat+SpaVar, stage 3), live events + both densities
            C3, H3, W3, D3 = STAGES[3]
            s3_bytes = 4.0 * B * H3 * W3 * (2 * C3 + 2 + 4)
            s3_ms = sum(e["s3_beg"].elapsed_time(e["s3_end"]) for e in ev) / len(ev)
            sparse, by_density = None, []
            if args.mask_density >= 1.0 and not args.no_density_sweep:
                (Lf, Rf) = hp.feats[3]
                for dens in (0.5, 0.3, 0.1, 0.05, 0.02):  # the kernel skips work ~ density^2
                    _, m2 = make_inputs(B, dev, dens, seed=4242)
                    t = time_kernel(lambda: hp.decnet.spamatvar_forward(Lf, Rf, m2[3][0], m2[3][1], D3,
                                                                        out=hp.outs[2]), 10)
                    # bytes: the algorithmic figure counts whole planes whatever the masks; the
                    # sparse-row kernel never fetches lines without an active pixel, so at low densities it moves
                    # less than that.  `achieved` uses min(algorithmic, PMC-counted) bytes per launch: the rate
                    # at which bytes really moved, never above the roofline by construction.
                    tr = traffic.get("spamat_fused_stage3_density_%.2f" % dens, {}).get("total_bytes")
                    tr = tr * (B / 8.0) if tr else None   # the PMC passes ran at 8 pairs per launch
                    used = min(s3_bytes, tr) if tr else s3_bytes
                    by_density.append({"mask_density": dens, "ms": t, "achieved": used / t / 1e6,
                                       "frac": used / t / 1e6 / HBM_PEAK_GBS, "traffic": tr,
                                       "algorithmic_bytes": s3_bytes, "bytes_counted_for_achieved": used})
                    # the same pass with the masks as the bit-packed copies decnet_detail_mask writes
                    # (decnet_spamatvar_forward_bits: identical outputs, 8 of the 88 bytes per pixel not read).
                    # frac_algorithmic: SURVEY 8d's byte count of the pass (float mask planes, what the reference
                    # interface moves) / time, capped at the peak; frac_moved: PMC-counted bytes / time.
                    rb, tb = pack_mask_bits(m2[3][0]), pack_mask_bits(m2[3][1])
                    tb_ms = time_kernel(lambda: hp.decnet.spamatvar_forward_bits(Lf, Rf, rb, tb, D3, out=hp.outs[2]), 10)
                    trb = traffic.get("spamat_fused_bits_stage3_density_%.2f" % dens, {}).get("total_bytes")
                    trb = trb * (B / 8.0) if trb else None
                    by_density[-1]["bit_masks"] = {
                        "ms": tb_ms, "traffic": trb,
                        "frac_algorithmic": min(1.0, s3_bytes / tb_ms / 1e6 / HBM_PEAK_GBS),
                        "frac_moved": (trb / tb_ms / 1e6 / HBM_PEAK_GBS) if trb else None}
                    if dens == 0.1:
                        sparse = {"bound": "hbm", "achieved": used / t / 1e6, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": used / t / 1e6 / HBM_PEAK_GBS, "traffic": tr,
                                  "kernel": "spamat sparse-row kernel (+ marker launch), fused fwd, stage 3",
                                  "mask_density": 0.1, "ms": t, "algorithmic_bytes": s3_bytes,
                                  "bytes_counted_for_achieved": used, "bit_masks": by_density[-1]["bit_masks"]}
            # the dense pass against the bound that applies to it: instruction issue (VALU + fp32 MFMA)
            valu = None
            if args.mask_density >= 1.0 and not args.no_valu_floor:
                fp32_dense = os.environ.get("DECNET_SPAMAT_DENSE", "") == "fp32"
                vf = valu_floor(B * H3 * ((W3 + 15) // 16), 30, 32 if fp32_dense else 8)
                if vf:
                    valu = {"bound": "valu", "achieved": vf["floor_ms"], "peak": s3_ms, "unit": "ms (floor / measured)",
                            "frac": vf["floor_ms"] / s3_ms, "traffic": None, "ms": s3_ms,
                            "kernel": "spamat fused fwd, stage 3, mask density 1.0", "detail": vf,
                            "note": "frac = issue floor of the kernel's own arithmetic / measured time: the softmax "
                                    "VALU passes of its 60 candidates per lane (microbenchmark, live) + the vector-issue "
                                    "cycles of its 30 MFMAs per wave-tile (%s)" % (
                                        "fp32 MFMA, 32 cycles each" if fp32_dense else "bf16x3 on v_mfma_f32_16x16x32_bf16, "
                                        "8 cycles each")}
        # Which roof the dominant kernel is under: the one it is CLOSER to.  Both fractions are in the object; "bound",
        # "achieved", "peak", "unit", "frac" are those of the larger one (bf16x3 Winograd GEMM at config 2: 340 MB of
        # algorithmic bytes in 0.11 ms = 0.38 of 8 TB/s against 0.31 of the dense bf16 matrix peak -> HBM).
        mfma_tf = gemm_mult * kern_flop / conv_ms / 1e9
        roof_mfma = {"achieved": mfma_tf, "peak": gemm_peak, "unit": "TFLOP/s", "frac": mfma_tf / gemm_peak}
        roof_hbm = None
        if kern_bytes:
            gbs = kern_bytes / conv_ms / 1e6
            roof_hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": kern_bytes}
        if roof_hbm and roof_hbm["frac"] > roof_mfma["frac"]:
            roof_head = dict(bound="hbm", achieved=roof_hbm["achieved"], peak=roof_hbm["peak"], unit=roof_hbm["unit"],
                             frac=roof_hbm["frac"], mfma=roof_mfma, hbm=roof_hbm)
        else:
            roof_head = dict(bound="mfma", achieved=roof_mfma["achieved"], peak=roof_mfma["peak"], unit=roof_mfma["unit"],
                             frac=roof_mfma["frac"], mfma=roof_mfma, hbm=roof_hbm)
        out = {
            "metric": ("stereo pairs/sec at 960x540x192disp" if args.config == 2 else
                       "stereo pairs/sec, BASELINE config %d shapes" % args.config) +
                      " (hot path: stage-0 dense + SpaMat/SpaVar stages 1-3)",
            "value": pairs / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (bf16x3 on the bf16 matrix pipe)" if gemm_mult > 1 else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE %s: batch=%d synthetic pairs per GPU, feature maps of the "
                                   "4-stage/scale-3 net, random-init CostRegNetNoDown(216)" % (CONFIG_NAME, B),
                       "pairs_per_gpu": B, "mask_density": args.mask_density,
                       "parallelism": "dp%d (pairs sharded, all_gather of disparity maps)" % world},
            # bf16x3 (default at Ci = 216): every fp32 product of the GEMM is executed as six bf16 products on
            # v_mfma_f32_16x16x32_bf16, so the kernel is priced with its EXECUTED flops (6 x the algorithmic ones)
            # against the dense bf16 peak; fp32_equivalent_tflops = algorithmic flops / time (157.3 would be the fp32 peak)
            "roofline": dict(roof_head, **{
                         # FROZEN in round 5 -- do not re-label between rounds
                         "definition": "dominant kernel = the launch with the largest share of the timed step (the "
                                       "Winograd GEMM, 7 launches per step).  hbm = its algorithmic operand bytes per "
                                       "launch (V in + M out + the weights it reads) / its live-measured launch time / "
                                       "8000 GB/s; mfma = its EXECUTED flops per launch (6 bf16 products per fp32 "
                                       "product for the bf16x3 kernel) / time / the dense peak of the pipe it runs on.  "
                                       "bound / achieved / peak / unit / frac are those of the roof with the LARGER "
                                       "fraction (the binding one); both sub-objects are always present.  traffic = "
                                       "PMC-counted bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md)",
                         "traffic": traffic.get(tkey, {}).get("total_bytes"),
                         "traffic_source": "profiles/traffic.json (committed rocprofv3 --pmc passes)", "kernel": kern_name,
                         "ms": conv_ms, "flop_per_launch": gemm_mult * kern_flop, "algorithmic_flop_per_launch": kern_flop,
                         "fp32_equivalent_tflops": kern_flop / conv_ms / 1e9,
                         "fp32_equivalent_frac_of_fp32_mfma_peak": kern_flop / conv_ms / 1e9 / MFMA_F32_PEAK_TF,
                         "arithmetic": gemm_arith, "stage0_ms_in_step": s0_ms,
                         # one of the seven 216 -> 216 layers: a seventh of the fused stack where the step runs it
                         # (conv3d_stack_ms), else one decnet_conv3d_wino_bn_act call (= conv3d_layer_unfused_ms)
                         "conv3d_layer_ms": layer_ms, "conv3d_stack_ms": stack_ms,
                         "conv3d_layer_unfused_ms": layer_unfused_ms,
                         "conv3d_layer_direct_equiv_tflops": conv_flop / layer_ms / 1e9}),
            "roofline_costvol": {"bound": "hbm", "achieved": s3_bytes / s3_ms / 1e6, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": s3_bytes / s3_ms / 1e6 / HBM_PEAK_GBS,
                                 "traffic": traffic.get("spamat_fused_stage3", {}).get("total_bytes"),
                                 "kernel": "spamat fused fwd, stage 3",
                                 "mask_density": args.mask_density, "ms": s3_ms,
                                 "bytes_per_launch": s3_bytes,
                                 "note": "at mask density 1.0 this pass is FP32-issue bound, not HBM bound: 806.7 M "
                                         "candidates x ~12 VALU-op equivalents + 8 MFMA MACs each cap it near 0.2 "
                                         "of 8 TB/s (DESIGN.md section 4); the HBM-shaped regime is the sparse one "
                                         "in roofline_costvol_sparse"},
        }
        out["preconditioning_steps"] = args.precondition_steps
        if blocks:
            out["value_blocks"] = blocks
        if smi:
            out["smi_under_load"] = smi
        if hp.coll:
            # the step's collective alone (not overlapped), and that what it delivers is what was sent
            with torch.no_grad():
                src = hp.outs3[0][0]
                t_ag = time_kernel(lambda: hp.dist.gather_disparity(src, n_pairs=world * B), 20, warm=3)
                got = hp.dist.gather_disparity(src, n_pairs=world * B)
                s_, e_ = hp.dist.shard_range(world * B, rank, world)
                out["collective"] = {"backend": torch.distributed.get_backend(), "world": world,
                                     "forced_at_world_1": world == 1, "all_gather_alone_ms": t_ag,
                                     "bytes_per_rank": src.numel() * 4,
                                     "own_shard_equals_sent": bool(torch.equal(got[s_:e_], src))}
        if valu:
            out["roofline_costvol_valu"] = valu
        if sparse:
            out["roofline_costvol_sparse"] = sparse
            out["roofline_costvol_sparse"]["by_density"] = by_density
            # the cost-volume pass against the HBM roof at the densities the reviews quote, one map (ALGORITHMIC bytes
            # of the pass -- whole planes, float masks, SURVEY 8d -- over the measured time, whatever really moved)
            fad = {"1.0": s3_bytes / s3_ms / 1e6 / HBM_PEAK_GBS}
            for row in by_density:
                if row["mask_density"] in (0.5, 0.3, 0.1):
                    fad["%.1f" % row["mask_density"]] = s3_bytes / row["ms"] / 1e6 / HBM_PEAK_GBS
            out["roofline_costvol"]["frac_at_density"] = fad
            # ... and inside `roofline`, the object the driver's record keeps: the cost-volume pass (stage 3, fused
            # SpaMat + SpaVar) against the HBM roof, beside the dominant kernel's own figures
            out["roofline"]["costvol_pass"] = {
                "kernel": "spamat fused fwd, stage 3", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "bytes_per_launch": s3_bytes, "frac_at_density": fad,
                "ms_at_density": dict({"1.0": s3_ms}, **{"%.1f" % r_["mask_density"]: r_["ms"] for r_ in by_density
                                                          if r_["mask_density"] in (0.5, 0.3, 0.1)}),
                "valu_floor_frac_dense": valu["frac"] if valu else None}
            out["roofline_costvol"]["ms_at_density"] = dict({"1.0": s3_ms}, **{
                "%.1f" % r_["mask_density"]: r_["ms"] for r_ in by_density if r_["mask_density"] in (0.5, 0.3, 0.1)})
        if world == 1 and not args.no_train and args.config == 2:
            try:
                out["train"] = train_leg(dev)
            except Exception as e:
                out["train"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if world == 1 and not args.no_e2e:
            try:
                out["e2e"] = e2e_bench(B, dev)
                # the end-to-end forward of the whole graph on the same batch (the HIP-graph replay where
                # the capture worked: same kernels, no launch gaps), beside `value` = the hot path alone
                hg = out["e2e"].get("hip_graph", {})
                out["value_end_to_end"] = max(out["e2e"]["value"], hg.get("value", 0.0))
                if not args.no_e2e_table:
                    out["e2e"]["roofline"] = e2e_kernel_table(B, out["e2e"].get("unit_work"))
                if not args.no_alt and os.environ.get("DECNET_CONV2D_ACC", "") == "":
                    out["e2e"]["two_accumulator_trunk"] = e2e_acc2_leg()
                if not args.no_cpu_baseline:
                    out["e2e"]["cpu_baseline"] = e2e_cpu_baseline()
            except Exception as e:                      # never lose the bench line to the extra leg
                out["e2e"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if world == 1 and not args.no_live_traffic:
            # roofline.traffic counted in this run (two rocprofv3 --pmc child passes); the committed figure stays on failure
            tb, note = live_traffic(args.config)
            if tb is not None:
                out["roofline"]["traffic_committed"] = out["roofline"]["traffic"]
                out["roofline"]["traffic"] = tb
                out["roofline"]["traffic_source"] = note
            else:
                out["roofline"]["traffic_source"] += "; live count not available (%s)" % note
        if (world == 1 and not args.no_alt and args.config == 2 and args.mask_density >= 1.0 and
                os.environ.get("DECNET_WINO_GEMM", "") == ""):
            out["alt_wino_gemm_fp32"] = alt_gemm_leg()
        if world == 1 and not args.no_alt:
            try:
                out["stage0_cost_funcs"] = cost_func_leg(hp.feats[0], dev)
            except Exception as e:                      # never lose the bench line to an extra leg
                out["stage0_cost_funcs"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(budget_s=args.cpu_budget)
        print(json.dumps(out), file=_JSON_OUT or sys.stdout, flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
