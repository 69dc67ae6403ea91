#!/usr/bin/env python
"""Benchmark of the hot path: training images/sec, CLIP ViT-B/32 + HSC, 224x224 (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the reference's inner loop (`src/eoe/training/ad_trainer.py:428-436`) over one synthetic
step batch of 128 normal + 128 OE already-normalised images per GPU: zero_grad -> encoder forward -> HSC loss ->
backward -> (gradient all-reduce) -> Adam -> anomaly scores from the pre-step features.  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line (contract in the task description), with two extra objects:
  roofline     -- the dominant kernel family's achieved TFLOP/s (algorithmic flops / hipEvent-measured launch time,
                  collected in a separate profiled pass inside this run) against the dense 16-bit MFMA peak;
  cpu_baseline -- the CPU oracle (a port of the reference step: oracle/) timed on this host's cores on a bounded
                  sample (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Kernel arguments in device memory (a HIP runtime setting, read when libamdhip64 is loaded, i.e. at `import torch`): by default the runtime
# keeps kernarg blocks in host memory and every workgroup's first scalar loads of its arguments cross PCIe -- a few microseconds per launch,
# 0.40 ms of a 10.46 ms ViT step with its ~250 launches (measured, round 4: 10.46 -> 10.06 ms, interleaved).  A default only: an explicit
# setting in the environment wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# multi-process GPU work on this pool needs dmabuf IPC (RCCL / cross-process tensors fail with "hipIpcGetMemHandle: invalid argument" otherwise);
# the boxes export it already -- a default only
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

MFMA_PEAK_TFLOPS = 2500.0    # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
F32_MFMA_PEAK_TFLOPS = 157.3 # fp32-input MFMA (v_mfma_f32_16x16x4_f32) = the fp32 vector rate (same guide)
HBM_PEAK_GBS = 8000.0
FWD_GFLOP_PER_IMG = 8.818    # ViT-B/32 image tower forward, 2 flop per MAC


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="normal images per GPU per step (an equal OE half is added)")
    ap.add_argument("--mode", choices=["full", "frozen", "eval"], default="full",
                    help="full fine-tune (config 5 style), frozen encoder + trained head (config 4), or eval = the forward-only "
                         "scoring loop of eval_cls (ad_trainer.py:473-550): encoder forward + anomaly scores, no backward / optimiser")
    ap.add_argument("--comm", choices=["auto", "native", "torch"], default="auto",
                    help="gradient transport under --gpus N: native = the C-ABI communicator (eoe_comm_*: RCCL on its own side HIP "
                         "stream, BatchNorm sums inside the library), torch = torch.distributed collectives; auto = native on the "
                         "nccl (RCCL) backend, torch otherwise (gloo rehearsals)")
    ap.add_argument("--comm-algo", choices=["rs_ag", "ring"], default="rs_ag",
                    help="native transport: reduce-scatter + all-gather per bucket (every xGMI link busy in both phases) or one all-reduce")
    ap.add_argument("--dtype", choices=["fp16", "bf16"], default="fp16")
    ap.add_argument("--bucket-dtype", choices=["fp32", "bf16"], default="fp32",
                    help="N > 1: the type the gradient buckets cross the ranks in (fp32 = the reference arithmetic; bf16 halves the xGMI bytes)")
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--model", choices=["vit", "cnn32", "wrn"], default="vit",
                    help="vit = the BASELINE.json metric config; cnn32 = secondary (config 1/2 backbone, 32x32); "
                         "wrn = secondary (WideResNet+CBAM, 224x224, config 3 backbone)")
    ap.add_argument("--nt-flags", type=int, default=None, help="tuning switch of the NT GEMM (A/B builds only)")
    ap.add_argument("--tn-flags", type=int, default=None, help="tuning switch of the wgrad GEMM")
    ap.add_argument("--grad-scale", type=float, default=None, help="loss-gradient scale (default: 256 for fp16, 1 otherwise)")
    ap.add_argument("--attn-flags", type=int, default=None, help="1: the one-wave attention backward kernel (A/B)")
    ap.add_argument("--side-stream", type=int, default=None, help="0: LayerNorm-1 backward after (not next to) the grouped wgrad")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="capture forward + loss + backward + scores of one step in a HIP graph and replay it (1 GPU only; "
                         "the optimiser step stays outside the graph).  auto = on for the launch-bound 32x32 configurations (cnn32, "
                         "wrn --res <= 64: ~400 launches of a few us each); the ViT / WideResNet-224 steps are GPU-bound and "
                         "measure the same either way")
    ap.add_argument("--parity-mode", action="store_true",
                    help="cnn32 / wrn: the convolutions and linear layers as exact-fp32 implicit GEMMs on the fp32 matrix cores "
                         "(v_mfma_f32_16x16x4_f32, csrc/parity.hip) -- the mode that holds the 1e-3 trajectory bar for the BatchNorm "
                         "encoders; peak 157 TFLOP/s")
    ap.add_argument("--conv-y16", action="store_true",
                    help="conv nets, fp16: keep the convolution outputs in fp16 in front of BatchNorm (ops.CONV_Y16; a speed option with one "
                         "more rounding point, see DESIGN.md section 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--serial-kernels", action="store_true",
                    help="no side-stream overlap of the weight-gradient launches anywhere (the per-kernel profiles under profiles/ are taken "
                         "this way: overlapped launches stretch each other's durations)")
    ap.add_argument("--res", type=int, default=None, help="input resolution of --model wrn (224 default; 32 = BASELINE.json config 2)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch normal (+ as many OE) images per GPU; strong: that many per JOB, split over the ranks")
    ap.add_argument("--no-torch-baseline", action="store_true")
    ap.add_argument("--no-box-probe", action="store_true", help="skip the three calibration probes in front of the timed region")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=5)
    return ap.parse_args()


def cpu_baseline(args):
    """the CPU oracle (port of the reference step) on a bounded sample: N=cpu_batch images, 2 warm-ups + cpu_steps (>= 5) timed steps
    """
    import torch
    from oracle import models as omodels, trainer as otrainer
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))   # the GPU box grants 16 host cores per GPU
    m = omodels.ClipViTNet(layers=args.layers, freeze=(args.mode == "frozen"))
    omodels.deterministic_init(m, tag="bench", layers=args.layers)
    nh = args.cpu_batch // 2
    batch = otrainer.synthetic_batch("bench/cpu", nh, nh, 224)
    if args.mode == "eval":
        otrainer.eval_scores(m, [batch] * 2, "hsc")                              # 2 warm-ups
        t0 = time.perf_counter()
        otrainer.eval_scores(m, [batch] * args.cpu_steps, "hsc")
        dt = time.perf_counter() - t0
    else:
        otrainer.train_steps(m, [batch] * 2, "hsc", lr=1e-4, weight_decay=1e-3)      # 2 warm-ups
        t0 = time.perf_counter()
        otrainer.train_steps(m, [batch] * args.cpu_steps, "hsc", lr=1e-4, weight_decay=1e-3)
        dt = time.perf_counter() - t0
    return {"value": round(args.cpu_steps * 2 * nh / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{args.cpu_steps} steps of {2 * nh} images after 2 warm-ups (ViT-B/32 {args.layers} layers, {args.mode}, fp32 oracle)"}


def torch_rocm_baseline(args, dev):
    """context beside the CPU baseline, NOT a target: the same step (same architecture, batch, objective, Adam) written with
    stock PyTorch-ROCm ops -- the oracle's modules moved to the GPU (GEMMs go to hipBLASLt / rocBLAS, everything else to eager
    elementwise kernels) with torch.optim.Adam, in fp32 and under fp16 autocast.  It answers "what would plain PyTorch do on this
    GPU"; the reference itself cannot travel to the GPU box."""
    import torch
    from oracle import models as omodels, objectives as oobj
    out = {"kind": "stock PyTorch-ROCm ops (oracle modules on the GPU, torch.optim.Adam), same step", "unit": "images/sec"}
    nb = args.batch
    imgs = torch.randn((2 * nb, 3, 224, 224), device=dev)
    lbls = torch.cat([torch.zeros(nb, dtype=torch.int64), torch.ones(nb, dtype=torch.int64)]).to(dev)
    for name, cast in (("fp32", None), ("fp16_autocast", torch.float16)):
        try:
            m = omodels.ClipViTNet(layers=args.layers, freeze=(args.mode == "frozen"))
            omodels.deterministic_init(m, tag="bench", layers=args.layers)
            m = m.to(dev).train()
            m.freeze_parts()
            opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-3)

            def one():
                opt.zero_grad()
                with torch.autocast("cuda", dtype=cast, enabled=cast is not None):
                    feats = m(imgs)
                loss = oobj.hsc_loss(feats.float(), lbls, 0)
                loss.backward()
                opt.step()
                return oobj.hsc_score(feats.detach().float())
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                one()
            torch.cuda.synchronize()
            out[name] = round(5 * 2 * nb / (time.perf_counter() - t0), 1)
            del m, opt
            torch.cuda.empty_cache()
        except Exception as e:                       # the baseline must never take the bench line down
            out[name] = None
            out[name + "_error"] = repr(e)[:200]
    return out


def box_probe(dev):
    """Three fixed probes of THIS box, run before the timed region (MI355X boxes of the pool differ by +-4 % on identical code): a bare
    fp16 MFMA loop (no memory traffic), a 1 GB 16-byte-per-lane copy, and this library's own 4096^3 NT GEMM.  The step time normalised
    to a reference box is ms_per_step * gemm_4096_tf / (the reference box's gemm_4096_tf) -- DESIGN.md section 5."""
    import ctypes as C
    import torch
    from eoe_amd import _lib, ops
    s = torch.cuda.current_stream().cuda_stream
    cus = torch.cuda.get_device_properties(dev).multi_processor_count

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps
            best = t if best is None else min(best, t)
        return best * 1e-3

    iters, blocks = 20000, cus * 8                      # 8 workgroups of 4 waves per CU = 8 waves per SIMD
    # (~15 ms per launch; 12 launches per timed repetition: the chip settles on its clock under load within the first -- MI355X_MICROARCH.md,
    #  "DVFS give-back" -- and the best of three repetitions is a steady-state figure)
    t = timed(lambda: _lib.check(_lib.lib.eoe_probe_mfma_f16(None, iters, blocks, s), "eoe_probe_mfma_f16"), 12)
    mfma_tf = 2.0 * 16 * 16 * 32 * 8 * iters * 4 * blocks / t / 1e12
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    t = timed(lambda: _lib.check(_lib.lib.eoe_probe_copy(dst.data_ptr(), src.data_ptr(), nbytes, s), "eoe_probe_copy"), 3)
    copy_gbs = 2.0 * nbytes / t / 1e9                   # bytes read + bytes written
    del src, dst
    a = torch.randn(4096, 4096, device=dev).half()
    b = (torch.randn(4096, 4096, device=dev) * 0.05).half()
    c = torch.empty(4096, 4096, device=dev, dtype=torch.float16)
    t = timed(lambda: ops.gemm_nt(a, b, c), 30)
    gemm_tf = 2.0 * 4096 ** 3 / t / 1e12
    del a, b, c
    # the step's own largest shape: c_fc forward, 12 800 x 3072 x 768 with the GELU pair of outputs (operands rotated through 4 sets: from HBM)
    sets = [(torch.randn(12800, 768, device=dev).half(), (torch.randn(3072, 768, device=dev) * 0.05).half(), torch.randn(3072, device=dev),
             torch.empty(12800, 3072, device=dev, dtype=torch.float16), torch.empty(12800, 3072, device=dev, dtype=torch.float16)) for _ in range(4)]

    def cfc():
        for a_, w_, bias_, out_, pre_ in sets:
            ops.gemm_nt(a_, w_, out_, bias=bias_, epilogue=ops.EPI_GELU, aux_out=pre_)
    t = timed(cfc, 8) / len(sets)
    cfc_tf = 2.0 * 12800 * 3072 * 768 / t / 1e12
    del sets
    torch.cuda.empty_cache()
    return {"mfma_f16_loop_tf": round(mfma_tf, 1), "hbm_copy_gbs": round(copy_gbs, 1), "gemm_4096_tf": round(gemm_tf, 1),
            "gemm_cfc_tf": round(cfc_tf, 1),
            "note": "bare v_mfma_f32_16x16x32_f16 loop (8 waves per SIMD); 1 GiB copy, read + written bytes; eoe_gemm_nt 4096^3 fp16; "
                    "eoe_gemm_nt 12800 x 3072 x 768 fp16 with the GELU pair (the step's largest shape)"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this same command (rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE in separate passes, tools/pmc_summary.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    PMC counters cannot be collected from inside the timed process, so this is the last profiled run, not this one."""
    import glob
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*", "bench_pmc_hbm_bytes*.csv")),
                   key=lambda q: (int("".join(c for c in os.path.basename(os.path.dirname(q)) if c.isdigit()) or 0), os.path.getmtime(q)))
    if not paths:
        return None, None
    tot = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}
    with open(paths[-1]) as f:
        for line in f:
            parts = line.strip().split(",")
            # (the library's scope names are prefixes of the kernel names; "cast" alone would also collect cast_transpose / cast_colsum)
            if len(parts) != 4 or parts[0] not in tot or (kernel + "_kernel" if kernel == "cast" else kernel) not in parts[1]:
                continue
            tot[parts[0]][0] += float(parts[3]) * 1024.0 * int(parts[2])
            tot[parts[0]][1] += int(parts[2])
    if not tot["FETCH_SIZE"][1] or not tot["WRITE_SIZE"][1]:
        return None, None
    per = 2.0 * tot["FETCH_SIZE"][0] / tot["FETCH_SIZE"][1] + tot["WRITE_SIZE"][0] / tot["WRITE_SIZE"][1]
    return round(per), os.path.relpath(paths[-1], os.path.dirname(os.path.abspath(__file__)))


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import eoe_amd
    from eoe_amd import _lib, parallel
    from eoe_amd.models import ClipViTB32Custom

    rank, world, local = parallel.init_from_env(os.environ.get("EOE_DIST_BACKEND", "nccl"))   # "nccl" = RCCL
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    eoe_amd.set_compute_dtype(args.dtype)
    if args.parity_mode:
        assert args.model in ("cnn32", "wrn"), "--parity-mode is for the BatchNorm encoders (the ViT meets the bar in its 16-bit mode)"
        eoe_amd.set_parity_mode(True)
    if args.conv_y16:
        assert args.model in ("cnn32", "wrn") and args.dtype == "fp16" and not args.parity_mode, "--conv-y16: conv nets, fp16 fast mode"
        from eoe_amd import ops as _opsy
        _opsy.CONV_Y16 = True
    # fp16: loss gradient x 256 against underflow in the 16-bit backward chain (FusedAdam divides it out)
    eoe_amd.set_grad_scale(args.grad_scale if args.grad_scale else eoe_amd.default_grad_scale())
    if args.tn_flags is not None:
        _lib.check(_lib.lib.eoe_set_option(b"tn_flags", args.tn_flags), "eoe_set_option")
    if args.side_stream is not None:
        _lib.check(_lib.lib.eoe_set_option(b"vit_side_stream", args.side_stream), "eoe_set_option")
    if args.serial_kernels:
        from eoe_amd import ops as _ops0
        _ops0.VIT_ASYNC_WGRAD = _ops0.CONV_ASYNC_WGRAD = False
        if args.side_stream is None:
            _lib.set_option("vit_side_stream", 0)          # every launch on one stream
    if args.nt_flags is not None:
        _lib.check(_lib.lib.eoe_set_option(b"nt_flags", args.nt_flags), "eoe_set_option")
    if args.attn_flags is not None:
        _lib.check(_lib.lib.eoe_set_option(b"attn_flags", args.attn_flags), "eoe_set_option")

    torch.manual_seed(0)
    res = 224
    if args.scaling == "strong":
        assert args.batch % world == 0, "--scaling strong needs --batch divisible by the number of ranks"
        args.batch //= world
    if args.model == "vit":
        model = ClipViTB32Custom(prediction_head=True, clf=False, freeze=(args.mode == "frozen"), layers=args.layers).to(dev).train()
        opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-3)     # train_clip_imagenet.py:13-14
        model.freeze_parts()
    elif args.model == "wrn":
        from eoe_amd.models import WideResNet
        res = args.res or 224
        model = WideResNet(res=res).to(dev).train()                                  # train_imagenet.py backbone (224); 32: config 2
        opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.0)
    else:
        from eoe_amd.models import CNN32
        res = 32
        model = CNN32(bias=True).to(dev).train()                                     # train_cifar.py:44
        opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.0)       # train_cifar.py:17-18
    # ---- the exchange step.  Default on RCCL: the library's own communicator -- gradient buckets reduce-scattered
    # and all-gathered on a side HIP stream from inside backward, BatchNorm sums added inside the library (no Python in that path).
    comm, comm_kind = parallel.make_comm(args.comm, args.comm_algo) if world > 1 else (None, "none")
    training = args.mode != "eval"
    arena = parallel.GradArena(model, comm=comm, bucket_dtype=torch.bfloat16 if args.bucket_dtype == "bf16" else None) if training else None
    if world > 1 and training:
        arena.install_hooks()
        if any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in model.modules()):
            parallel.enable_sync_bn(comm=comm)   # global-batch BatchNorm statistics, as the single-device reference computes them

    nb = args.batch
    n_local = 2 * nb
    n_global = n_local * world
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    imgs = torch.randn((n_local, 3, res, res), generator=gen, device=dev)
    imgs[nb:] += 0.5 * torch.randn((1, 3, res, res), generator=torch.Generator(device=dev).manual_seed(7), device=dev)
    lbls = torch.cat([torch.zeros(nb, dtype=torch.int64), torch.ones(nb, dtype=torch.int64)]).to(dev)
    score_buf = torch.empty((args.steps + args.warmup + 8, n_local), dtype=torch.float32, device=dev)

    wait_events = []                 # (before, after) events around the wait for the step's collectives: the EXPOSED communication

    def step(i):
        opt.zero_grad()
        feats = model(imgs)
        loss = eoe_amd.hsc_loss(feats, lbls, 0, 1.0 / n_global)
        loss.backward()
        if world > 1:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            arena.finish()
            b.record()
            wait_events.append((a, b))
        else:
            arena.finish()
        opt.step()
        score_buf[i % score_buf.shape[0]] = eoe_amd.hsc_score(feats)
        return loss

    if not training:
        model.eval()

        def step(i):                                       # noqa: F811  (eval_cls: forward-only scoring, ad_trainer.py:498-512)
            with torch.no_grad():
                feats = model(imgs)
            score_buf[i % score_buf.shape[0]] = eoe_amd.hsc_score(feats)
            return score_buf[i % score_buf.shape[0]][0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    launch_bound = args.model == "cnn32" or (args.model == "wrn" and res <= 64)
    use_graph = training and (args.graph == "on" or (args.graph == "auto" and launch_bound and world == 1))
    eager_step = step
    if use_graph:
        assert world == 1, "--graph on is a single-GPU option"
        graphed = eoe_amd.GraphedStep(model, lambda f, y: eoe_amd.hsc_loss(f, y, 0, 1.0 / n_global), eoe_amd.hsc_score, imgs, lbls)

        def step(i):                                       # noqa: F811  (replaces the eager step)
            opt.zero_grad()
            loss, scores = graphed(imgs, lbls)
            opt.step()
            score_buf[i % score_buf.shape[0]] = scores
            return loss

    # (every rank runs the probes -- rank 0's numbers go into the line --: the ranks reach the first collective of the warm-up together)
    box = box_probe(dev) if not args.no_box_probe else None
    for i in range(args.warmup):
        loss = step(i)
    sync()
    wait_events.clear()
    host_s = 0.0                                           # host time inside step() (enqueue; no device synchronisation in there)
    t0 = time.perf_counter()
    for i in range(args.steps):
        h0 = time.perf_counter()
        loss = step(args.warmup + i)
        host_s += time.perf_counter() - h0
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    comm_exposed_ms = None
    if wait_events:
        comm_exposed_ms = sum(a.elapsed_time(b) for a, b in wait_events) / len(wait_events)
    final_loss = loss.item()
    if final_loss != final_loss:                         # diagnose before failing: which tensors went non-finite
        import sys as _sys
        bad_p = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
        bad_g = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        bad_b = [n for n, b in model.named_buffers() if b.dtype.is_floating_point and not torch.isfinite(b).all()]
        first = [i for i in range(score_buf.shape[0]) if not torch.isfinite(score_buf[i]).all()][:3]
        print(f"NaN loss: non-finite params {bad_p[:6]} grads {bad_g[:6]} buffers {bad_b[:6]}; score rows non-finite from {first}",
              file=_sys.stderr, flush=True)
    assert final_loss == final_loss, "NaN loss"
    # epoch-tail metric of the reference (ad_trainer.py:456-471: ROC-AUC of the scores collected during training), on the
    # last timed step's scores of this rank (outside the timed region)
    from eoe_amd import metrics
    last = score_buf[(args.warmup + args.steps - 1) % score_buf.shape[0]].float().cpu().numpy()
    auc = float(metrics.roc_auc(lbls.cpu().numpy(), last))

    roof = None
    if use_graph:
        step = eager_step                                  # the per-kernel profile runs the eager launches
    if not args.no_roofline:
        # separate profiled pass: hipEvents around every kernel launch (inside the library, on the launch stream).  The weight-gradient
        # launches are SERIALISED for this pass: in the timed region they run on a side stream under the next block's / layer's kernels
        # (ops.VIT_ASYNC_WGRAD, ops.CONV_ASYNC_WGRAD), which stretches the durations of whatever shares the chip with them -- a kernel's
        # launch duration only means "its own time" when it runs alone.  `value` / `ms_per_step` are measured with the overlap.
        from eoe_amd import ops as _ops
        overlap = (_ops.VIT_ASYNC_WGRAD, _ops.CONV_ASYNC_WGRAD)
        _ops.VIT_ASYNC_WGRAD = _ops.CONV_ASYNC_WGRAD = False
        # ... and LayerNorm-1 backward is kept off the library's side stream (vit.cpp runs it NEXT TO the grouped wgrad otherwise: both
        # launches' durations then count the time they share -- BENCH_r03's layernorm_bwd read 2.4 ms per step for 0.84 of work)
        side = _lib.get_option("vit_side_stream")
        _lib.set_option("vit_side_stream", 0)
        _lib.prof_enable(True)
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        _lib.set_option("vit_side_stream", side)
        _ops.VIT_ASYNC_WGRAD, _ops.CONV_ASYNC_WGRAD = overlap
        prof = _lib.prof_collect()
        gemm = {k: v for k, v in prof.items() if k.startswith("gemm") or k.startswith("conv_f32")}
        tot_ms = sum(v["total_ms"] for v in prof.values())
        dom = max(gemm, key=lambda k: gemm[k]["total_ms"])
        d = gemm[dom]
        achieved = d["flops"] / (d["total_ms"] * 1e-3) / 1e12
        headline = args.model == "vit" and args.mode == "full" and args.dtype == "fp16"
        traffic, traffic_src = pmc_traffic(dom) if headline else (None, None)
        peak = F32_MFMA_PEAK_TFLOPS if dom.startswith("conv_f32") else MFMA_PEAK_TFLOPS
        # the HBM-bound kernels of the step against the 8 TB/s roofline: algorithmic bytes per launch (the library's own count) over the
        # launch's hipEvent duration; `pmc_bytes` = 2 x FETCH_SIZE + WRITE_SIZE per launch from the committed counter pass
        hbm = []
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
            if k in gemm or not v["bytes"] or not v["launches"] or v["total_ms"] <= 0:
                continue
            gbs = v["bytes"] / (v["total_ms"] * 1e-3) / 1e9
            hbm.append({"kernel": k, "launches_per_step": v["launches"] // 3, "avg_us": round(v["total_ms"] * 1e3 / v["launches"], 2),
                        "bytes_per_launch": round(v["bytes"] / v["launches"]), "pmc_bytes_per_launch": pmc_traffic(k)[0] if headline else None,
                        "GB/s": round(gbs, 1), "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4)})
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 1), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                # the same against what THIS box's matrix pipes deliver in a bare fp16 MFMA loop (box.mfma_f16_loop_tf: the clock the chip holds
                # under an MFMA-dense loop is about half the nominal 2.4 GHz the 2.5 PF figure assumes)
                "frac_of_box_mfma_loop": round(achieved / box["mfma_f16_loop_tf"], 4) if (box and not dom.startswith("conv_f32")) else None,
                "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE)", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                "avg_launch_us": round(d["total_ms"] * 1e3 / d["launches"], 2), "launches_per_step": d["launches"] // 3,
                "kernels_ms_per_step": {k: round(v["total_ms"] / 3, 3) for k, v in sorted(prof.items())},
                "profiled_ms_per_step": round(tot_ms / 3, 3),
                "hbm": hbm,
                "pass": "3 extra steps, hipEvents per launch, every launch on ONE stream (in the timed steps the weight-gradient launches "
                        "overlap the dgrad chain and LayerNorm-1 backward runs beside them on a side stream)" if training
                        else "3 extra steps, hipEvents per launch"}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = n_global * args.steps / elapsed
        flop_per_img = FWD_GFLOP_PER_IMG * (3.0 if args.mode == "full" else 1.0) * args.layers / 12.0
        # the last block runs on its class-token rows only (ln_post reads nothing else): the out-projection and MLP GEMMs of the other 49
        # tokens -- 0.520 GFLOP of the tower's forward -- are not executed and not counted
        from eoe_amd import ops as _opsc
        cls_only = args.model == "vit" and _opsc.VIT_CLS_ONLY_LAST
        if cls_only:
            flop_per_img -= 0.520 * (3.0 if args.mode == "full" else 1.0)
        mode_txt = {"full": "full fine-tune", "frozen": "frozen encoder", "eval": "forward-only scoring (eval_cls)"}[args.mode]
        if args.model == "cnn32":
            flop_per_img = 0.179
        if args.model == "wrn":
            flop_per_img = 10.89 * (res / 224.0) ** 2                                # 3 x 3.63 GFLOP at 224
        workload = {
            "vit": (f"CLIP ViT-B/32 ({args.layers} layers) + Linear(512,256) + HSC, "
                    f"{mode_txt}, Adam lr 1e-4 wd 1e-3, "
                    f"224x224, {nb} normal + {nb} OE images per GPU per step"),
            "cnn32": f"CNN32(bias=True) + HSC, Adam lr 1e-3, 32x32, {nb} normal + {nb} OE images per GPU per step",
            "wrn": f"WideResNet(ResNet-18 + CBAM) + HSC, Adam lr 1e-3, {res}x{res}, {nb} normal + {nb} OE images per GPU per step",
        }[args.model]
        verb = "train" if training else "eval"
        metric = {"vit": f"{verb} images/sec, CLIP ViT-B/32 + HSC, 224x224", "cnn32": f"{verb} images/sec, CNN32 + HSC, 32x32",
                  "wrn": f"{verb} images/sec, WideResNet+CBAM + HSC, {res}x{res}"}[args.model]
        if not training:
            flop_per_img = {"vit": FWD_GFLOP_PER_IMG * args.layers / 12.0 - (0.520 if cls_only else 0.0), "cnn32": 0.0597,
                            "wrn": 3.63 * (res / 224.0) ** 2}[args.model]
        out = {
            "metric": metric,
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32" if args.parity_mode else args.dtype, "data": "synthetic",
            "config": {"workload": workload, "global_batch": n_global, "parallelism": f"dp{world}",
                       "launch": "hip graph replay" if use_graph else ("eager, kernels serialised" if args.serial_kernels else
                                                                        "eager, weight-gradient launches on a side stream under the dgrad chain"
                                                                        if training else "eager"),
                       "arithmetic": "exact fp32 (fp32 MFMA convolutions / linears)" if args.parity_mode else
                                     ("16-bit MFMA operands, fp32 accumulate" + (", fp16 convolution outputs in front of BatchNorm" if args.conv_y16 else "")),
                       **({"last_block": "class-token rows only past the attention (what ln_post reads; same embedding and gradients; "
                                         "EOE_VIT_CLS_ONLY=0 computes the unread rows too)"} if cls_only else {})},
            "model_tflops": round(value * flop_per_img / 1e3, 1),
            "mfma_roofline_frac_end_to_end": round(value * flop_per_img / 1e3 / (MFMA_PEAK_TFLOPS * world), 4),
            "final_loss": round(final_loss, 5) if training else None, "auc_last_step": round(auc, 4),
            "host_enqueue_ms": round(host_s / args.steps * 1e3, 3),
        }
        if box is not None:
            out["box"] = box
        if world > 1 and training:
            # what the first hardware scaling run needs to be read: how much was sent, in how many collectives, and how long the
            # compute stream stood waiting for them at the end of backward (hipEvents around the join; everything else overlapped)
            sent = [hi - lo for _, lo, hi in arena.block_buckets] + [hi - lo for _, lo, hi in arena.run_buckets]
            esz = 2 if args.bucket_dtype == "bf16" else 4
            out["comm"] = {"transport": comm_kind, "buckets": len(sent), "bucket_dtype": args.bucket_dtype, "allreduce_bytes_per_step": int(esz * sum(sent)),
                           "largest_bucket_bytes": int(esz * max(sent)), "comm_exposed_ms": round(comm_exposed_ms, 3),
                           "sync_bn": bool(parallel._bn_sync_cb is not None)}
        if roof is not None:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline and args.model == "vit":
            out["cpu_baseline"] = cpu_baseline(args)
        if world == 1 and not args.no_torch_baseline and args.model == "vit" and training:
            del model, opt, arena
            torch.cuda.empty_cache()
            out["torch_rocm_baseline"] = torch_rocm_baseline(args, dev)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if training:
            parallel.disable_sync_bn()
            arena.remove_hooks()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
