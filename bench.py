#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the escape-time hot path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one frame of BASELINE.json's metric configuration:
Mandelbrot 4096x4096, max_iter 1024, fp64, default viewport (configs[1], "C2"), output planes
resident in HBM.  metric = Mpixels/s (whole job).  At N > 1 every frame is cut into row strips dealt
round-robin to the ranks; the K frames are processed in groups of N, frame g*N + j being gathered to
rank j by one RCCL all-to-all per group (rotating roots, fractalrenderer_amd/distributed.py
FrameExchange), double-buffered so that the exchange of group g overlaps the rendering of group g+1.
The ranks ship the 8-byte smooth-count plane and the destination recolours it (bit-identical to a
direct render), so every frame ends complete -- RGBA f32 + nu -- in the HBM of one GPU, as at N = 1.
Strong scaling: the same K frames whatever N.

Launching.  The driver starts N > 1 as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(WORLD_SIZE set: this process IS a rank).  From a bare shell `python bench.py --gpus N` works too: with WORLD_SIZE
unset and N > 1 the process becomes a LAUNCHER -- before torch is imported or anything touches the GPU it starts
`python -m torch.distributed.run` as a child process (never an exec), relays rank 0's JSON line and exits with the
child's status.

--mode sequence (default) times K frames in rotating-root groups as above; --mode frame times the north-star's literal
case at N > 1: ONE frame tiled over the N GPUs as row strips with a final gather to rank 0 (StripGather: grouped RCCL
send/recv into the root, double-buffered so that gather(n) overlaps render(n+1)); at N = 1 both modes are the same
whole-frame render.

--host node: the C-ABI multi-GPU host instead of torch.distributed -- ONE process, ctypes -> fr_node over devices 0..N-1
(one worker thread per device, frame slots, render lanes; fractalrenderer_amd/csrc/fr_node.cpp), from a bare shell:
`python bench.py --gpus N --host node`.  Same JSON contract; value = --mode sequence (frames in flight, rotating roots) or
--mode frame (one frame at a time, gathered to device 0) with the automatic gather, and a "node" object with every
(gather, mode) pair that the box allows, each verified bitwise against fr_render.  --node-parts P cuts every frame into P
parts per device (one-card rehearsals of the band arithmetic: --gpus 1 --node-parts 4 = devices {0,0,0,0}).
Under the driver's torch.distributed.run command (WORLD_SIZE set, N > 1) the torch path stays the headline and rank 0 adds
an informational "fr_node" object, measured in-process after the timed region while the other ranks wait at a host-side
(gloo) barrier; FR_BENCH_NODE_LEG=0 switches that leg off.

Order of the N = 1 legs: the informational `periodicity` leg (the library's default, cycle closing on) runs first, the
headline leg behind it -- each is W untimed + K timed steps of the full frame; a GPU that has been idle takes tens of
milliseconds of load to reach its sustained clocks (tools/clock_ramp.py), and a 20-step headline leg right behind the
process's first kernel would time that ramp.  The line says so (`leg_order`).  The render contexts run with the library's
default `timing` = off (no HIP event pair around every render): the timed region is measured with events of its own.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline       the metric's own roofline ("achieved HBM GB/s vs peak"): algorithmic bytes
                 16 B/pixel (RGBA f32, write-once) / average kernel time measured with HIP events
                 on the launch stream over the timed region; traffic = PMC WRITE_SIZE+FETCH_SIZE per
                 launch from the committed rocprofv3 pass (profiles/), or null.
  roofline_valu  the bound that actually limits the kernel: fp64 VALU issue.
                 achieved / frac = 6 VALU instructions x executed iterations (counted exactly from the iter
                 plane) / kernel time over peak = 39.3 T lane-instructions/s (78.6 TFLOP/s FMA-counted / 2:
                 the loop cannot contract) -- the instruction-issue fraction, which cannot exceed 1;
                 survey_8d = the same with SURVEY.md's 8 as-written flops per iteration (= frac x 8/6).
  cpu_baseline   the CPU oracle (oracle/fr_oracle.c, OpenMP) timed on a bounded sample of the same frame, rank 0,
                 N = 1 only: on 16 threads (the box's CPU share per GPU), on one thread and on every hardware thread
                 the process can see.  kind "port": the reference has no CPU path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (fractal, precision, W, H, state kwargs)
    "c1": dict(desc="C1 mandelbrot 512x512 max_iter=256 fp64 default viewport (BASELINE.json configs[0], the CPU-runnable case)",
               fractal="Mandelbrot", precision="F64", W=512, H=512, cpu_rows=512, cpu_passes=20, state=dict(max_iterations=256)),
    "c2": dict(desc="C2 mandelbrot 4096x4096 max_iter=1024 fp64 default viewport (center -0.5,0 zoom 3.0)",
               fractal="Mandelbrot", precision="F64", W=4096, H=4096, cpu_rows=4096, state=dict(max_iterations=1024)),
    "c2_reset": dict(desc="mandelbrot 4096x4096 max_iter=1024 fp64 reset() viewport (zoom 1.5)",
                     fractal="Mandelbrot", precision="F64", W=4096, H=4096, state=dict(max_iterations=1024, zoom=1.5)),
    "c3": dict(desc="C3 julia c=-0.8+0.156i 4096x4096 max_iter=2048 fp32 centre (0,0) zoom 3.0",
               fractal="JuliaSet", precision="F32", W=4096, H=4096, cpu_rows=4096,
               state=dict(max_iterations=2048, center_x=0.0, center_y=0.0, julia_c_real=-0.8, julia_c_imag=0.156)),
    "c4": dict(desc="C4 deep-zoom mandelbrot 8192x8192 max_iter=16384 fp64 seahorse zoom 1e-6",
               fractal="Mandelbrot", precision="F64", W=8192, H=8192, cpu_rows=64,
               state=dict(max_iterations=16384, center_x=-0.743643887037151, center_y=0.13182590420533, zoom=1e-6)),
    "c5": dict(desc="C5 mandelbrot 8192x8192 max_iter=4096 fp64 seahorse zoom 0.008 (one .franim keyframe view)",
               fractal="Mandelbrot", precision="F64", W=8192, H=8192, cpu_rows=64,
               state=dict(max_iterations=4096, center_x=-0.743643887037151, center_y=0.13182590420533, zoom=0.008)),
    "rabbit": dict(desc="julia c=-0.123+0.745i (Douady rabbit: attracting 3-cycle, filled interior) 4096x4096 max_iter=2048 fp32",
                   fractal="JuliaSet", precision="F32", W=4096, H=4096, cpu_rows=512,
                   state=dict(max_iterations=2048, center_x=0.0, center_y=0.0, julia_c_real=-0.123, julia_c_imag=0.745)),
    "hd": dict(desc="interactive size: mandelbrot 1920x1080 max_iter=256 fp32 default viewport (the reference's draw loop)",
               fractal="Mandelbrot", precision="F32", W=1920, H=1080, cpu_rows=1080, state=dict(max_iterations=256)),
    "hd1k": dict(desc="interactive size: mandelbrot 1920x1080 max_iter=1024 fp64 default viewport",
                 fractal="Mandelbrot", precision="F64", W=1920, H=1080, cpu_rows=1080, state=dict(max_iterations=1024)),
    "uhd": dict(desc="interactive size: mandelbrot 3840x2160 max_iter=256 fp32 default viewport",
                fractal="Mandelbrot", precision="F32", W=3840, H=2160, cpu_rows=2160, state=dict(max_iterations=256)),
    "uhd1k": dict(desc="interactive size: mandelbrot 3840x2160 max_iter=1024 fp64 default viewport",
                  fractal="Mandelbrot", precision="F64", W=3840, H=2160, cpu_rows=2160, state=dict(max_iterations=1024)),
    # the other kernels of the path (each with its own bound; profiles/r02_*): the reference's perturbation shader, the
    # effects variant of the tile kernel, and -- `kernel` entries, no render in the timed region -- the recolour and the
    # two export kernels of the frame-output path
    "deepzoom": dict(desc="Deep_Zoom (shaders/test_deep_zoom.comp: fp32 perturbation against an fp64 reference orbit) 4096x4096 "
                          "max_iter=2000 seahorse zoom 1e-6",
                     fractal="Deep_Zoom", precision="F32", W=4096, H=4096, cpu_rows=256, flops_per_update=23,
                     state=dict(max_iterations=2000, center_x=-0.743643887037151, center_y=0.13182590420533, zoom=1e-6,
                                use_perturbation=True)),
    "trap": dict(desc="orbit trap blend: mandelbrot 4096x4096 max_iter=1024 fp64 default viewport "
                      "(shaders/mandelbrot.comp:163-166,193-198; lean tile pass + lane pool in their code-3 instantiations: the "
                      "trap's minimum is the constant 0 as the shader is written)",
                 fractal="Mandelbrot", precision="F64", W=4096, H=4096, cpu_rows=512,
                 state=dict(max_iterations=1024, orbit_trap_enabled=True)),
    "stripes": dict(desc="stripe shading: mandelbrot 4096x4096 max_iter=1024 fp64 default viewport "
                         "(shaders/mandelbrot.comp:201-205; lean tile pass + lane pool in their stripe instantiations)",
                    fractal="Mandelbrot", precision="F64", W=4096, H=4096, cpu_rows=512,
                    state=dict(max_iterations=1024, stripe_enabled=True)),
    "colorize": dict(desc="fr_colorize_async: colour from the fp64 smooth-count plane of the C2 frame, 4096x4096 "
                          "(8 B read + 16 B written per pixel)", kernel="colorize", bytes_per_pixel=24,
                     fractal="Mandelbrot", precision="F64", W=4096, H=4096, state=dict(max_iterations=1024)),
    "export8": dict(desc="fr_export_rgb8: RGBA f32 -> RGB8 (fp16 rounding, ACES, gamma, truncation, flip; "
                         "src/vk_engine.cpp:1344-1371), 4096x4096 (16 B read + 3 B written per pixel)",
                    kernel="export8", bytes_per_pixel=19,
                    fractal="Mandelbrot", precision="F64", W=4096, H=4096, state=dict(max_iterations=1024)),
    "export16": dict(desc="fr_export_rgb16: RGBA f32 -> RGB16 (clamp, truncation, flip; src/vk_engine.cpp:2054-2073), 8192x8192 "
                          "(16 B read + 6 B written per pixel)", kernel="export16", bytes_per_pixel=22,
                     fractal="Mandelbrot", precision="F64", W=8192, H=8192, state=dict(max_iterations=256)),
    # diagnostic only (tools/timeline.py, tools/sweep_opts.py): every pixel escapes at i <= 1 -- the per-pixel skeleton
    "far": dict(desc="diagnostic: far-exterior view, mandelbrot 4096x4096 max_iter=1024 fp64 centre (8,8) zoom 2",
                fractal="Mandelbrot", precision="F64", W=4096, H=4096, cpu_rows=4096,
                state=dict(max_iterations=1024, center_x=8.0, center_y=8.0, zoom=2.0)),
}

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TOPS = 39.3     # 78.6 TFLOP/s vector fp64 counts FMA as 2; contraction-off loop issues 1 op/slot
FP32_VALU_PEAK_TOPS = 78.6     # 157.3 TFLOP/s / 2


def cpu_baseline(workload: dict) -> dict:
    """Oracle timed on the host cores on evenly spaced row bands of the same frame."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    w = workload
    st = w["state"]
    fmap = {"Mandelbrot": 0, "JuliaSet": 1, "BurningShip": 2, "Deep_Zoom": 5}
    p = O.OracleParams(fractal=fmap[w["fractal"]], precision=1 if w["precision"] == "F64" else 0,
                       **{k: (int(v) if isinstance(v, bool) else v) for k, v in st.items()})
    W, H = w["W"], w["H"]
    bands = 16
    band_rows = max(1, w.get("cpu_rows", 1024) // bands)      # sized for ~10-30 core-seconds of CPU work
    # the GPU box gives one GPU a 16-CPU share of its host: use that many threads (override: FR_CPU_THREADS)
    threads = min(O.max_threads(), int(os.environ.get("FR_CPU_THREADS", "16")))
    O.render(p, W, H, y0=0, y1=2, threads=threads, planes=False)          # warm the thread pool
    passes = int(w.get("cpu_passes", 3))                      # ~10-30 core-seconds in total
    t0 = time.perf_counter()
    px = 0
    for _ in range(passes):
        for b in range(bands):
            y0 = (H // bands) * b + (H // bands - band_rows) // 2
            O.render(p, W, H, y0=y0, y1=y0 + band_rows, threads=threads, planes=False)
            px += band_rows * W
    dt = time.perf_counter() - t0
    # SURVEY.md section 8d also asks for the single-thread figure: one pass over narrower bands of the same frame
    rows1 = max(1, band_rows // 4)
    t1s = time.perf_counter()
    px1 = 0
    for b in range(bands):
        y0 = (H // bands) * b + (H // bands - rows1) // 2
        O.render(p, W, H, y0=y0, y1=y0 + rows1, threads=1, planes=False)
        px1 += rows1 * W
    dt1 = time.perf_counter() - t1s
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # the reference's only fp64 z <- z^2 + c (DeepZoomManager::compute_reference_orbit, src/deep_zoom_system.cpp:378-424,
    # which prints this rate itself, :356-359): the restated loop at an interior point, one thread
    n_orbit = 1 << 22
    buf = np.zeros((n_orbit, 2), np.float64)
    t1 = time.perf_counter()
    got = O.lib().fro_reference_orbit(-0.5, 0.0, n_orbit, buf.ctypes.data)
    orbit_rate = got / (time.perf_counter() - t1)
    # (after the single-thread timings: 128 idle OpenMP workers spinning after their region slow a lone thread down)
    # ... and beside the 16-thread figure (the box's CPU share per GPU) the one on every hardware thread the process can see
    # (one pass over the same bands; on a box whose cgroup caps the process at its share this cannot be faster, and says so)
    all_threads = O.max_threads()
    O.render(p, W, H, y0=0, y1=2, threads=all_threads, planes=False)
    tas = time.perf_counter()
    pxa = 0
    for b in range(bands):
        y0 = (H // bands) * b + (H // bands - band_rows) // 2
        O.render(p, W, H, y0=y0, y1=y0 + band_rows, threads=all_threads, planes=False)
        pxa += band_rows * W
    dta = time.perf_counter() - tas
    return {"value": round(px / dt / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "single_thread": {"value": round(px1 / dt1 / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
                              "sample": f"1 pass over {bands} evenly spaced bands of {rows1} rows ({px1} pixels), {dt1:.1f} s"},
            "all_hardware_threads": {"value": round(pxa / dta / 1e6, 3), "unit": "Mpixels/s", "cores": all_threads,
                                     "sample": f"1 pass over the same bands ({pxa} pixels) with {all_threads} OpenMP threads, {dta:.1f} s"},
            "reference_orbit_iterations_per_s": round(orbit_rate, 0),
            "sample": f"{passes} passes over {bands} evenly spaced bands of {band_rows} rows ({px // passes} of {W*H} pixels) of the same frame, "
                      f"oracle/fr_oracle.c -O2 -ffp-contract=off, OpenMP schedule(dynamic,1), {dt:.1f} s",
            "cpu": cpu, "nproc": os.cpu_count()}


def pmc_traffic(workload_name: str):
    """HBM bytes per launch from the committed rocprofv3 --pmc pass (profiles/pmc_traffic.json)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
        e = d.get(workload_name)
        return e.get("hbm_bytes_per_launch") if e else None
    except (OSError, ValueError):
        return None


def pmc_valu_busy(workload_name: str):
    """VALU busy of the frame's kernels from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json):
    SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); None for workloads that were not profiled."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        e = json.load(open(path)).get(workload_name) or {}
        vb = e.get("valu_busy")
        return {"frame": round(vb["frame"], 4), "per_kernel": {k: round(v, 4) for k, v in vb["per_kernel"].items()},
                "sustained_clock_ghz": round(vb["clock_ghz"], 3), "source": e.get("source")} if vb else None
    except (OSError, ValueError, KeyError, TypeError):
        return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="sequence", choices=["sequence", "frame"],
                    help="N > 1: 'sequence' = K frames in rotating-root groups (FrameExchange); 'frame' = the north-star's "
                         "literal case, every frame tiled over the N GPUs as row strips and gathered to rank 0 (StripGather)")
    ap.add_argument("--host", default="torch", choices=["torch", "node"],
                    help="'torch' = one process per GPU, torch.distributed (the driver's command); 'node' = ONE process driving "
                         "all N GPUs through the C ABI's fr_node (bare shell only)")
    ap.add_argument("--node-parts", type=int, default=1, help="--host node: parts per device (each device ordinal listed this often)")
    ap.add_argument("--node-slots", type=int, default=0, help="--host node: frames in flight (0: 2, or N at N > 2)")
    ap.add_argument("--node-lanes", type=int, default=0, help="--host node: render contexts per part (0: 2)")
    ap.add_argument("--gather", default="auto", choices=["auto", "peer", "rccl"], help="--host node: the gather of the headline value")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipelined", action="store_true",
                    help="N = 1: after the timed region, also time the same K steps with two frames in flight "
                         "(informational `pipelined` object; off by default so that a rocprofv3 trace of the "
                         "default command only contains the one-frame-at-a-time launches the roofline is quoted on)")
    ap.add_argument("--no-periodicity", action="store_true",
                    help="skip the informational leg with the library's exact cycle closing switched on")
    ap.add_argument("--options", default="", help="fr_ctx_set_option pairs for every render context, 'name=value,name=value' "
                                                  "(A/B runs and profiles of non-default schedules)")
    ap.add_argument("--wg-per-cu", type=int, default=0)
    ap.add_argument("--run-max", type=int, default=0)
    ap.add_argument("--shape", type=int, default=0)
    ap.add_argument("--rows-per-strip", type=int, default=0)
    ap.add_argument("--payload", default="auto", choices=["auto", "nu", "rgba"],
                    help="N > 1: plane shipped over xGMI (auto: nu when the colour is a function of nu)")
    ap.add_argument("--layout", default="bands", choices=["bands", "strips"],
                    help="N > 1: contiguous row bands that rotate over the frames of a group and are received in place "
                         "(default), or row strips interleaved inside every frame")
    ap.add_argument("--render-lanes", type=int, default=0,
                    help="N > 1: concurrent render contexts/streams per rank (0: min(4, N))")
    ap.add_argument("--dist-backend", default="nccl", help="rehearsals only: gloo + --same-device on one card")
    ap.add_argument("--same-device", action="store_true", help="rehearsals only: every rank uses cuda:0")
    ap.add_argument("--cpu-rehearsal", action="store_true",
                    help="N > 1, no GPU: gloo ranks run the launcher, the rendezvous and the exchange of --mode on CPU tensors "
                         "filled with a stand-in pattern instead of rendered planes (tests/test_dist_cpu.py drives the "
                         "launcher path with it); the printed value is NOT a measurement")
    return ap.parse_args(argv)


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` from a bare shell (WORLD_SIZE unset, N > 1): start the N ranks as CHILD processes through
    torch.distributed.run -- before this process has imported torch or touched the GPU, and without exec (a process that
    has initialised the GPU must never be replaced) -- relay rank 0's JSON line, return the children's status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL peer mappings need it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:                                 # stderr goes straight through; stdout is scanned for THE line
        t = out.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py launcher: the ranks exited cleanly but rank 0 printed no JSON line\n")
        rc = 1
    return rc


def cpu_rehearsal(args) -> None:
    """The N > 1 control flow of --mode without a GPU: gloo rendezvous, the exchange on CPU tensors whose 'rendered' planes
    are a stand-in pattern (frame * 65536 + global row), every delivered frame verified; rank 0 prints a JSON line shaped
    like the real one.  Exercised by tests/test_dist_cpu.py through the launcher; never a measurement."""
    import torch
    import torch.distributed as dist
    from fractalrenderer_amd.distributed import FrameExchange, StripGather
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("FR_BENCH_REHEARSAL_FAIL_RANK") == str(rank):
        raise SystemExit(3)                                  # the launcher must hand a rank's failure on
    dist.init_process_group("gloo")
    W, H = 64, 48
    cpu = torch.device("cpu")
    pattern = lambda frame, rows: (frame * 65536.0 + torch.from_numpy(rows).double())[:, None].expand(len(rows), W)  # noqa: E731
    whole = lambda frame: pattern(frame, __import__("numpy").arange(H))  # noqa: E731
    ok = True
    t0 = time.perf_counter()
    if args.mode == "frame":
        sg = StripGather(W, H, 1, torch.float64, cpu, rows_per_strip=args.rows_per_strip)

        def render_fn(shard, out, frame):
            out.copy_(pattern(frame, shard.global_rows(H)).unsqueeze(-1))

        for f in range(args.warmup + args.steps):
            slot = sg.submit(render_fn, f)
            sg.drain()
            if rank == 0:
                ok = ok and bool(torch.equal(sg.frames[slot][..., 0], whole(f)))
        desc = f"StripGather to rank 0, strips of {sg.R} rows"
    else:
        fx = FrameExchange(W, H, payload="nu", nu_dtype=torch.float64, device=cpu, rows_per_strip=args.rows_per_strip,
                           layout=args.layout)

        def render_fn(shard, out, frame, plane, lane=0):
            out.copy_(pattern(frame, shard.global_rows(H)))

        def colorize_fn(nu_frame, rgba_frame, frame):
            rgba_frame.copy_(nu_frame.float().unsqueeze(-1).expand(H, W, 4))

        fx.prime()
        f = 0
        while f < args.warmup + args.steps:
            count = min(world, args.warmup + args.steps - f)
            slot = fx.submit_group(render_fn, f, count, colorize_fn)
            fx.drain()
            if rank < count:
                ok = ok and bool(torch.equal(fx.frame_nu[slot], whole(f + rank)))
            f += count
        desc = f"FrameExchange, {'bands' if fx.bands else 'strips'} of {fx.R} rows"
    dt = time.perf_counter() - t0
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        n = args.warmup + args.steps
        print(json.dumps({"metric": "rehearsal (no GPU): launcher + rendezvous + exchange on CPU tensors; not a measurement",
                          "value": round(n * W * H / dt / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(dt / n * 1e3, 4), "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "stand-in pattern",
                          "rehearsal": "cpu", "exchange_verified": bool(flag.item()),
                          "config": {"workload": f"{W}x{H} stand-in planes", "mode": args.mode, "parallelism": desc}}), flush=True)
    dist.destroy_process_group()
    if not flag.item():
        raise SystemExit(4)


def kernel_workload(args, w) -> None:
    """--workload colorize | export8 | export16: one step = ONE launch of that kernel over a resident plane (the render
    that produces its input is outside the timed region).  HBM-bound kernels: roofline = algorithmic bytes / HIP-event time."""
    import numpy as np
    import torch
    import fractalrenderer_amd as fr
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    dev = torch.device("cuda", 0)
    W, H = w["W"], w["H"]
    r = fr.Renderer(0)
    state = fr.FractalState(**w["state"])
    kind = w["kernel"]
    rgba = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
    nu = torch.empty((H, W), dtype=torch.float64, device=dev)
    r.render(state, W, H, rgba=rgba, nu=nu, post_chain=(kind != "colorize"))
    stream = torch.cuda.Stream(device=dev)
    h = stream.cuda_stream
    if kind == "colorize":
        out = torch.empty_like(rgba)
        step = lambda: r.colorize(state, nu, out, stream=h)                                   # noqa: E731
    elif kind == "export8":
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        step = lambda: r.export_rgb8(rgba, W, H, out=out, through_half=True, stream=h)        # noqa: E731
    else:
        out = torch.empty((H, W, 3), dtype=torch.int16, device=dev)
        step = lambda: r.export_rgb16(rgba, W, H, out=out, stream=h)                          # noqa: E731
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    verified_against = "the synchronous entry point"
    if kind == "colorize":                      # the recoloured plane IS the rendered one
        verified = bool(torch.equal(out, rgba))
        verified_against = "the rendered colour plane"
    elif kind == "export8" and not args.no_cpu_baseline:
        # the checker: the restated CPU loop of src/vk_engine.cpp:1344-1371 over the WHOLE plane -- the bytes must be identical
        from oracle import oracle as O
        O.build()
        verified = bool(np.array_equal(out.cpu().numpy(), O.export_rgb8(rgba.cpu().numpy(), through_half=True)))
        verified_against = "oracle.export_rgb8 (byte identity over the whole plane)"
    else:
        ref = r.export_rgb8(rgba, W, H, through_half=True) if kind == "export8" else r.export_rgb16(rgba, W, H)
        verified = bool(torch.equal(out, ref))
    bpp = w["bytes_per_pixel"]
    gbs = bpp * W * H / (kernel_ms * 1e-3) / 1e9
    line = {"metric": f"Mpixels/s ({args.workload})", "value": round(args.steps * W * H / dt / 1e6, 2), "unit": "Mpixels/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["desc"], "compute_units": r.compute_units},
            "output_verified": verified, "output_verified_against": verified_against,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                         "traffic": pmc_traffic(args.workload), "kernel_ms": round(kernel_ms, 4),
                         "note": f"algorithmic bytes {bpp} B/pixel; the guide's measured streaming ceiling is ~6.3 TB/s (79 % of spec)"}}
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        O.build()
        rows = 512
        src = rgba[:rows].cpu().numpy()
        t1 = time.perf_counter()
        if kind == "colorize":
            O.colorize(O.OracleParams(max_iterations=state.max_iterations), nu[:rows].cpu().numpy())
        else:
            O.export_rgb8(src, through_half=True)     # the restated CPU loop of src/vk_engine.cpp:1344-1371 (8-bit form for both)
        dtc = time.perf_counter() - t1
        line["cpu_baseline"] = {"value": round(rows * W / dtc / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                                "sample": f"{rows} rows of the same plane, oracle (single thread), {dtc:.2f} s"}
    print(json.dumps(line), flush=True)
    r.close()



def node_measure(fr, torch, w, devices, steps, warmup, mode, gather, slots, lanes, options="", loopback=False) -> dict:
    """K frames of workload `w` through fr_node over `devices` (ordinals, may repeat), device planes on each frame's root.
    mode "sequence": up to `slots` frames in flight on `lanes` render contexts per part, frame f gathered to root f % n;
    mode "frame": one frame at a time, every frame gathered to part 0.  Wall time between device-wide synchronisations;
    afterwards one frame per root is compared bitwise with fr_render on that root's device."""
    W, H = w["W"], w["H"]
    state = fr.FractalState(**w["state"])
    ftype, prec = fr.FractalType[w["fractal"]], fr.Precision[w["precision"]]
    n = len(devices)
    g = {"auto": fr._capi.FR_GATHER_AUTO, "peer": fr._capi.FR_GATHER_PEER, "rccl": fr._capi.FR_GATHER_RCCL}[gather]
    node = fr.Node(devices)
    try:
        node.set_option("periodicity", -1)                     # the reference's iteration count, as the headline
        for kv in filter(None, options.split(",")):
            k, v = kv.split("=")
            node.set_option(k.strip(), int(v, 0))
        node.set_option("slots", slots if mode == "sequence" else 1)
        node.set_option("lanes", lanes if mode == "sequence" else 1)
        if loopback:
            node.set_tuning("rccl_loopback", 1)
        node.set_tuning("rccl_timeout_ms", 10000)
        node.set_option("gather", g)
        nplanes = n * ((slots + n - 1) // n) if mode == "sequence" else 1
        planes = [torch.empty((H, W, 4), dtype=torch.float32, device=torch.device("cuda", devices[i % n])) for i in range(nplanes)]

        def sync_all():
            for d in sorted(set(devices)):
                torch.cuda.synchronize(d)

        def run(count):
            if mode == "frame":
                for _ in range(count):
                    node.render(state, W, H, root=0, fractal_type=ftype, precision=prec, rgba=planes[0])
            else:
                for f in range(count):
                    node.submit(state, W, H, root=f % n, fractal_type=ftype, precision=prec, rgba=planes[f % nplanes])
                node.wait()

        run(max(warmup, nplanes))
        sync_all()
        t0 = time.perf_counter()
        run(steps)
        sync_all()
        dt = time.perf_counter() - t0
        used = node.last_gather()
        # verification, outside the timed region: the frame on every root against fr_render there (with the event pair on:
        # the parts' kernel times of that frame go into the line)
        node.set_option("timing", 1)
        ok = True
        roots = range(n) if mode == "sequence" else [0]
        for root in roots:
            dev = torch.device("cuda", devices[root])
            got = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
            want = torch.empty_like(got)
            torch.cuda.synchronize(dev)
            node.render(state, W, H, root=root, fractal_type=ftype, precision=prec, rgba=got)
            with fr.Renderer(devices[root]) as rr:
                rr.set_option("periodicity", -1)
                rr.render(state, W, H, fractal_type=ftype, precision=prec, rgba=want)
            ok = ok and bool(torch.equal(got, want))
            del got, want
            if len(set(devices)) == 1:
                break                                          # one card: every root is the same memory
        part_ms = [round(node.last_kernel_ms(k), 4) for k in range(n)]
        return {"value": round(steps * W * H / dt / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(dt / steps * 1e3, 4),
                "mode": mode, "gather": {fr._capi.FR_GATHER_PEER: "peer", fr._capi.FR_GATHER_RCCL: "rccl"}.get(used, str(used)),
                "gather_asked": gather, "parts": n, "devices": sorted(set(devices)), "slots": slots if mode == "sequence" else 1,
                "lanes": lanes if mode == "sequence" else 1, "last_part_kernel_ms": part_ms, "exchange_verified": ok}
    finally:
        node.close()


def node_matrix(fr, torch, w, args, n_gpus) -> dict:
    """Every (gather, mode) pair the box allows, each its own fr_node; failures are recorded, not raised."""
    devices = [d for d in range(n_gpus) for _ in range(max(1, args.node_parts))]
    if args.same_device:
        devices = [0] * len(devices)
    distinct = len(set(devices)) == len(devices)
    slots = args.node_slots or (2 if len(devices) <= 2 else min(8, len(devices)))
    lanes = args.node_lanes or 2
    gathers = ["peer"] + (["rccl"] if (distinct and len(devices) > 1) else [])
    out = {"devices": devices, "runs": []}
    for g in gathers:
        for mode in ("sequence", "frame"):
            try:
                out["runs"].append(node_measure(fr, torch, w, devices, args.steps, args.warmup, mode, g, slots, lanes, args.options))
            except Exception as e:  # noqa: BLE001  (an informational leg must not take the line down)
                out["runs"].append({"mode": mode, "gather_asked": g, "error": str(e)[:400]})
    if len(devices) == 1:
        # one card, one part: the whole RCCL frame path through a one-rank communicator (tuning "rccl_loopback")
        try:
            out["runs"].append(dict(node_measure(fr, torch, w, devices, args.steps, args.warmup, "sequence", "rccl", slots, lanes,
                                                 args.options, loopback=True), note="rccl_loopback: one-rank communicator, strips sent to self"))
        except Exception as e:  # noqa: BLE001
            out["runs"].append({"mode": "sequence", "gather_asked": "rccl (loopback)", "error": str(e)[:400]})
    return out


def node_host(args) -> None:
    """`python bench.py --gpus N --host node`: ONE process, fr_node over devices 0..N-1 through ctypes."""
    import torch
    import fractalrenderer_amd as fr
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    have = torch.cuda.device_count()
    if args.gpus > have and not args.same_device:
        raise SystemExit(f"--gpus {args.gpus} but this process sees {have} device(s) (--same-device rehearses on cuda:0)")
    w = WORKLOADS[args.workload]
    W, H = w["W"], w["H"]
    m = node_matrix(fr, torch, w, args, args.gpus)
    want_g = "peer" if args.gather == "auto" else args.gather
    head = next((r for r in m["runs"] if r.get("mode") == args.mode and r.get("gather_asked") == want_g and "value" in r), None)
    if head is None:
        head = next((r for r in m["runs"] if "value" in r), None)
    if head is None:
        raise SystemExit("fr_node: no run succeeded: " + json.dumps(m))
    gbs = 16 * W * H / (head["ms_per_step"] * 1e-3) / 1e9
    prec64 = w["precision"] == "F64"
    line = {"metric": "Mpixels/s at 4096x4096 max_iter=1024; achieved HBM GB/s vs peak" if args.workload == "c2"
                      else f"Mpixels/s ({args.workload})",
            "value": head["value"], "unit": "Mpixels/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if prec64 else "f32", "data": "synthetic", "host": "node",
            "config": {"workload": w["desc"], "output": "RGBA f32 linear colour, 16 B/pixel, resident in HBM on the frame's root",
                       "mode": head["mode"],
                       "parallelism": f"ONE process, C ABI fr_node: {head['parts']} part(s) on device(s) {head['devices']}, one worker thread "
                                      f"per part, row strips dealt round-robin, gather = {head['gather']}, {head['slots']} frame slot(s), "
                                      f"{head['lanes']} render lane(s) per part" + (", roots rotating" if head["mode"] == "sequence" else ", root 0")},
            "exchange_verified": all(r.get("exchange_verified", False) for r in m["runs"] if "value" in r),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                         "traffic": None, "kernel_ms": None,
                         "note": "16 B/pixel over the WALL time per frame of this host (frames overlap: no per-launch duration); the "
                                 "kernels are VALU-bound, see the default command's roofline_valu"},
            "cpu_baseline": None,
            "node": m}
    print(json.dumps(line), flush=True)


def main() -> None:
    argv = sys.argv[1:]
    args = parse_args(argv)
    # before torch is imported and before any HIP call: dmabuf IPC for RCCL's peer mappings (read at runtime init)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.host == "node":
        if "WORLD_SIZE" in os.environ:
            raise SystemExit("--host node is ONE process: start it from a bare shell, not under torch.distributed.run")
        if "kernel" in WORKLOADS[args.workload]:
            raise SystemExit("kernel workloads are single-context")
        node_host(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, argv))
    if args.cpu_rehearsal:
        if args.gpus < 2:
            raise SystemExit("--cpu-rehearsal rehearses the N > 1 path: use --gpus 2 or more")
        cpu_rehearsal(args)
        return

    if "kernel" in WORKLOADS[args.workload]:
        if args.gpus != 1:
            raise SystemExit("kernel workloads are single-GPU")
        kernel_workload(args, WORKLOADS[args.workload])
        return

    import numpy as np
    import torch
    import torch.distributed as dist

    import fractalrenderer_amd as fr
    from fractalrenderer_amd.distributed import FrameExchange, StripGather

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    host_group = None
    node_leg = world > 1 and os.environ.get("FR_BENCH_NODE_LEG", "1") != "0"
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
            if node_leg:
                host_group = dist.new_group(backend="gloo")   # a HOST-side barrier for the fr_node leg: an RCCL barrier would
                                                              # keep a spinning kernel on every waiting rank's GPU
        else:
            dist.init_process_group(args.dist_backend)

    w = WORKLOADS[args.workload]
    W, H = w["W"], w["H"]
    state = fr.FractalState(**w["state"])
    ftype = fr.FractalType[w["fractal"]]
    prec = fr.Precision[w["precision"]]
    def new_renderer():
        rr = fr.Renderer(local_rank, timing=False)      # the C library's default: no event pair around every render (the timed
                                                         # region is measured with events of its own around the K steps)
        rr.set_tuning(args.wg_per_cu, args.run_max, args.shape)
        # The library closes cycles by default ("periodicity": exact, planes byte-identical, but fewer iterations than the
        # reference's shaders execute).  Every timed leg except the `periodicity` one runs with it OFF, so that `value`,
        # `roofline` and `roofline_valu` are quoted on the reference's iteration count.
        rr.set_option("periodicity", -1)
        for kv in filter(None, args.options.split(",")):
            k, v = kv.split("=")
            rr.set_option(k.strip(), int(v, 0))
        return rr

    r = new_renderer()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms = None
    executed = None
    if world == 1:
        rgba = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
        # exact executed-iteration count of this frame from the iter plane (outside the timed region)
        it = torch.empty((H, W), dtype=torch.int32, device=dev)
        r.render(state, W, H, fractal_type=ftype, precision=prec, rgba=rgba, iter=it)
        mi = state.max_iterations
        executed = int(torch.where(it < mi, it.to(torch.int64) + 1, torch.full_like(it, mi, dtype=torch.int64)).sum())
        del it
        stream = torch.cuda.Stream(device=dev)   # a real (non-null) HIP stream: launches AND events go here
        h = stream.cuda_stream
        assert h != 0

        def step(_k):
            r.render(state, W, H, fractal_type=ftype, precision=prec, rgba=rgba, sync=False, stream=h)

        # ORDER OF THE LEGS.  The informational `periodicity` leg runs FIRST, the headline leg after it: a GPU that has been idle
        # takes tens of milliseconds of load to reach its sustained clocks (tools/clock_ramp.py: the first 20 C2 frames of a
        # process average 0.78 ms, frames 50 on 0.68 ms), and with the driver's --steps 20 --warmup 5 a headline leg timed
        # right behind the process's first kernel measures that ramp, not the kernels.  Both legs are the contract's W untimed +
        # K timed steps of the full workload; nothing is added or skipped, only their order is chosen ("leg_order" in the line).
        # Secondary, informational: the same K steps with the library's default, cycle closing ("periodicity") ON.  The lane pool
        # then retires an orbit as interior the moment it returns to an earlier state of its own (it can never
        # escape: the update is a deterministic function of (z, c)); every plane stays byte-identical
        # (tests/test_gpu_parity.py::test_periodicity_never_changes_a_pixel) but FEWER iterations are executed
        # than the reference's shaders would run, so it is NOT the headline `value` and carries no roofline.
        dt_cyc = None
        check = None
        if not args.no_periodicity:
            r.set_option("periodicity", 0)             # the library's default
            check = torch.empty_like(rgba)
            r.render(state, W, H, fractal_type=ftype, precision=prec, rgba=check)
            for k in range(args.warmup):
                step(k)
            barrier()
            t0c = time.perf_counter()
            for k in range(args.steps):
                step(k)
            barrier()
            dt_cyc = time.perf_counter() - t0c
            r.set_option("periodicity", -1)

        # the headline leg
        for k in range(args.warmup):
            step(k)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)                       # HIP events on the launch stream, over the timed region
        for k in range(args.steps):
            step(k)
        ev1.record(stream)
        barrier()
        dt = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / args.steps
        if check is not None:
            cyc_identical = bool(torch.equal(check, rgba))
            del check

        # Secondary, informational: the same K steps with TWO frames in flight (a second render context on a
        # second stream, frames alternating), which is how an animation export runs (distributed.py).  It fills
        # the phase boundaries of a frame (tile-pass tail, launch gaps, pool-pass drain) with the other frame's
        # work.  NOT the headline `value`: that stays one frame at a time, so that `roofline` keeps its
        # per-launch meaning and agrees with the rocprofv3 kernel durations.
        dt_pipe = None
        if args.pipelined:
            r2 = new_renderer()
            rgba2 = torch.empty_like(rgba)
            stream2 = torch.cuda.Stream(device=dev)
            pair = ((r, rgba, h), (r2, rgba2, stream2.cuda_stream))

            def step2(k):
                rr, buf, hh = pair[k & 1]
                rr.render(state, W, H, fractal_type=ftype, precision=prec, rgba=buf, sync=False, stream=hh)

            for k in range(max(2, args.warmup)):
                step2(k)
            barrier()
            t0p = time.perf_counter()
            for k in range(args.steps):
                step2(k)
            barrier()
            dt_pipe = time.perf_counter() - t0p
            r2.close()
            del rgba2
    elif args.mode == "frame":
        # The north-star's literal case: ONE frame tiled over the N GPUs, gathered to rank 0.  Interleaved row strips
        # (contiguous H/N bands would leave the ranks that hold the set with several times the work), one render per
        # rank and frame, one grouped batch of RCCL send/recv into the root (N-1 concurrent xGMI transfers),
        # double-buffered: gather(n) overlaps render(n+1).  Payload nu (8 / 4 B per pixel, recoloured on the root,
        # bit-identical) where the colour is a function of nu, else the 16-byte colour plane.
        payload = args.payload
        if payload == "auto":
            payload = "nu" if r.colorize_supported(state, ftype, prec) else "rgba"
        nu_dtype = torch.float64 if prec == fr.Precision.F64 else torch.float32
        lanes = 1
        ctxs = [r]
        fx = sg = StripGather(W, H, 4 if payload == "rgba" else 1, torch.float32 if payload == "rgba" else nu_dtype, dev,
                              rows_per_strip=args.rows_per_strip)
        root_rgba = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) for _ in range(sg.nbuf)] \
            if (rank == 0 and payload == "nu") else []

        def render_fn(shard, out, _frame):
            kw = {"nu": out} if payload == "nu" else {"rgba": out}
            r.render(state, W, H, fractal_type=ftype, precision=prec, shard=shard, sync=False,
                     stream=torch.cuda.current_stream().cuda_stream, **kw)

        def run(nframes):
            for f in range(nframes):
                slot = sg.submit(render_fn, f)
                if rank == 0 and payload == "nu":                # recolour the assembled frame behind its gather
                    with torch.cuda.stream(sg.comm):
                        r.colorize(state, sg.frames[slot][..., 0], root_rgba[slot], fractal_type=ftype, precision=prec,
                                   stream=torch.cuda.current_stream().cuda_stream)
                        sg.gathered[slot].record(sg.comm)
            sg.drain()

        run(max(1, args.warmup))                                 # also opens the peer connections
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        r.set_option("timing", 1)                                # (off in the timed region: the library's default)
        run(1)                                                   # outside the timed region: verify the gathered frame
        last_ms = r.last_kernel_ms()
        lane_ms = [last_ms]
        okv = 1
        if rank == 0:
            direct = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
            r.render(state, W, H, fractal_type=ftype, precision=prec, rgba=direct)
            got = root_rgba[0] if payload == "nu" else sg.frames[0]
            okv = 1 if torch.equal(got, direct) else 0
            del direct
        ok = torch.tensor([okv], dtype=torch.int32, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        exchange_verified = bool(ok.item())
    else:
        payload = args.payload
        if payload == "auto":
            payload = "nu" if r.colorize_supported(state, ftype, prec) else "rgba"
        nu_dtype = torch.float64 if prec == fr.Precision.F64 else torch.float32
        lanes = args.render_lanes or min(4, world)
        fx = FrameExchange(W, H, payload=payload, nu_dtype=nu_dtype, device=dev, rows_per_strip=args.rows_per_strip,
                           render_lanes=lanes, layout=args.layout)
        # one render context per lane: a context is not re-entrant, distinct contexts run concurrently
        ctxs = [r] + [new_renderer() for _ in range(lanes - 1)]

        def render_fn(shard, out, _frame, plane, lane=0):
            kw = {"nu": out} if plane == "nu" else {"rgba": out}
            ctxs[lane].render(state, W, H, fractal_type=ftype, precision=prec, shard=shard, sync=False,
                              stream=torch.cuda.current_stream().cuda_stream, **kw)

        def colorize_fn(nu_frame, rgba_frame, _frame):
            r.colorize(state, nu_frame, rgba_frame, fractal_type=ftype, precision=prec,
                       stream=torch.cuda.current_stream().cuda_stream)

        def run(nframes):
            f = 0
            while f < nframes:                       # groups of `world` frames; the last one may be partial
                count = min(world, nframes - f)
                fx.submit_group(render_fn, f, count, colorize_fn if payload == "nu" else None)
                f += count
            fx.drain()

        fx.prime()                                   # peer connections are set up outside the timed region
        run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        # outside the timed region: one more group, and every rank compares the frame it was handed with a direct
        # single-GPU render of the same frame (bitwise); the verdict of all ranks goes into the JSON line
        for c in ctxs:
            c.set_option("timing", 1)                    # (off in the timed region: the library's default)
        slot = fx.submit_group(render_fn, 0, world, colorize_fn if payload == "nu" else None)
        fx.drain()
        last_ms = r.last_kernel_ms()
        lane_ms = [c.last_kernel_ms() for c in ctxs]     # device time of each render context's last launch
        direct = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
        r.render(state, W, H, fractal_type=ftype, precision=prec, rgba=direct)
        ok = torch.tensor([1 if torch.equal(fx.frame_rgba[slot], direct) else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        exchange_verified = bool(ok.item())
        del direct

    # MAX over ranks
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Informational, after the timed region: the same frames through the C ABI's multi-GPU host (fr_node: ONE process --
    # this one, rank 0 -- driving all N devices, worker thread per device, frame slots, render lanes, both gathers), while
    # the other ranks wait at a host-side barrier with idle GPUs.  Never the headline; failures are recorded, not raised.
    fr_node_obj = None
    node_leg_stuck = False
    if node_leg:
        torch.cuda.synchronize()
        dist.barrier(group=host_group)
        if rank == 0:
            leg = argparse.Namespace(**vars(args))
            leg.steps, leg.warmup = min(args.steps, 200), min(max(args.warmup, 2), 20)
            # on a worker thread with a deadline: this leg has never run on more than one card, and a hang in it (a
            # communicator that does not come up) must not take the measured headline with it
            import threading
            box = {}

            def run_leg():
                try:
                    if not args.same_device and torch.cuda.device_count() < world:
                        raise RuntimeError(f"rank 0 sees {torch.cuda.device_count()} device(s), needs {world}")
                    m = node_matrix(fr, torch, w, leg, world)
                    m["note"] = ("informational: ONE process (rank 0) drives all devices through the C ABI's fr_node after the "
                                 "timed region, the other ranks idle at a gloo barrier; per run: K frames, wall time, verified "
                                 "bitwise against fr_render on every root")
                    box["obj"] = m
                except Exception as e:  # noqa: BLE001
                    box["obj"] = {"error": str(e)[:400]}

            th = threading.Thread(target=run_leg, daemon=True)
            th.start()
            th.join(float(os.environ.get("FR_BENCH_NODE_LEG_TIMEOUT", "120")))
            node_leg_stuck = th.is_alive()
            fr_node_obj = {"error": "the fr_node leg did not finish within its deadline (FR_BENCH_NODE_LEG_TIMEOUT)"} if node_leg_stuck \
                else box.get("obj", {"error": "the fr_node leg ended without a result"})
        dist.barrier(group=host_group)

    if rank == 0:
        mpx = args.steps * W * H / dt / 1e6
        out = {
            "metric": "Mpixels/s at 4096x4096 max_iter=1024; achieved HBM GB/s vs peak" if args.workload == "c2"
                      else f"Mpixels/s ({args.workload})",
            "value": round(mpx, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None, "dtype": "f64" if prec == fr.Precision.F64 else "f32", "data": "synthetic",
            "config": {"workload": w["desc"], "output": "RGBA f32 linear colour, 16 B/pixel, resident in HBM",
                       "parallelism": "1 GPU, persistent tile queue" if world == 1 else
                                      (f"mode frame: every frame tiled over {world} GPUs as row strips of {fx.R} rows dealt round-robin, "
                                       f"gathered to rank 0 by one grouped RCCL send/recv batch per frame (payload: {payload} plane"
                                       f"{', recoloured on the root' if payload == 'nu' else ''}; double-buffered: gather(n) overlaps "
                                       f"render(n+1))" if args.mode == "frame" else
                                       (f"disjoint row bands of {fx.R} rows, one per GPU and frame, the band of a GPU rotating over "
                                        f"the frames of a group; " if fx.bands else
                                        f"row strips of {fx.R} rows round-robin over {world} GPUs; ") +
                                       f"frames in groups of {world}, "
                                       f"frame g*{world}+j gathered to rank j by one RCCL all-to-all per group "
                                       f"(payload: {payload} plane, {'recoloured at the destination, ' if payload == 'nu' else ''}"
                                       f"double-buffered, {lanes} concurrent render contexts per rank)"),
                       "mode": args.mode if world > 1 else "frame",
                       "compute_units": r.compute_units},
        }
        if world == 1:
            bytes_per_launch = 16 * W * H
            gbs = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(args.workload),
                               "kernel_ms": round(kernel_ms, 4),
                               "note": "the metric's HBM roofline; the kernel is fp64-VALU-bound, see roofline_valu"}
            peak = FP64_VALU_PEAK_TOPS if prec == fr.Precision.F64 else FP32_VALU_PEAK_TOPS
            fpu = float(w.get("flops_per_update", 8))
            tops = fpu * executed / (kernel_ms * 1e-3) / 1e12
            if fpu == 8:
                # PRIMARY: instruction issue.  An update is 6 VALU instructions (two of them FMAs by exact powers of two, each
                # standing for two of the 8 as-written operations, bit for bit), so 6 x executed iterations over the non-FMA
                # issue peak is a fraction that cannot exceed 1.  SURVEY.md section 8d's counting -- 8 as-written flops per
                # iteration over the same peak -- is kept beside it as survey_8d; it flatters by 8/6 and can exceed 1.
                tins = 6.0 * executed / (kernel_ms * 1e-3) / 1e12
                out["roofline_valu"] = {"bound": "valu_" + out["dtype"], "achieved": round(tins, 3), "peak": peak,
                                        "unit": "T lane-instructions/s (6 VALU instructions per executed iteration; peak = one "
                                                "non-FMA instruction per lane and cycle)",
                                        "frac": round(tins / peak, 4),
                                        "survey_8d": {"achieved": round(tops, 3), "unit": "Tflop/s (8 as-written flops per executed "
                                                      "iteration, no FMA credit)", "frac": round(tops / peak, 4),
                                                      "note": "SURVEY.md section 8d's counting; = frac x 8/6, can exceed 1"},
                                        "issue_frac": round(tins / peak, 4),
                                        "valu_busy_pmc": pmc_valu_busy(args.workload),
                                        "executed_iterations": executed,
                                        "mean_iterations_per_pixel": round(executed / (W * H), 2)}
            else:
                out["roofline_valu"] = {"bound": "valu_" + out["dtype"], "achieved": round(tops, 3), "peak": peak,
                                        "unit": f"Tflop/s ({fpu:g} flop per executed iteration, no FMA credit)",
                                        "frac": round(tops / peak, 4),
                                        "issue_note": "as-written operations of one perturbed update, shaders/test_deep_zoom.comp:153-173: "
                                                      "2 complex products with the reference point and delta (6 + 5), three additions of "
                                                      "pairs (6), z = ref + delta (2), dot(z, z) (3), compare (1)",
                                        "valu_busy_pmc": pmc_valu_busy(args.workload),
                                        "executed_iterations": executed,
                                        "mean_iterations_per_pixel": round(executed / (W * H), 2)}
            if dt_cyc is not None:
                out["leg_order"] = ["periodicity (informational; it also takes the cold GPU's clock ramp)", "headline"]
                out["periodicity"] = {"value": round(args.steps * W * H / dt_cyc / 1e6, 2), "unit": "Mpixels/s",
                                      "ms_per_step": round(dt_cyc / args.steps * 1e3, 4),
                                      "output_identical_to_headline_run": cyc_identical,
                                      "note": "informational: the library's DEFAULT (cycle closing on) -- orbits that return to an "
                                              "earlier state of their own are retired as interior at once (exact; planes "
                                              "byte-identical), so fewer iterations run than the reference executes; the "
                                              "headline value above switches it off (fr_ctx_set_option(\"periodicity\", -1)) "
                                              "and iterates every interior sample to max_iter"}
            if dt_pipe is not None:
                out["pipelined"] = {"frames_in_flight": 2, "value": round(args.steps * W * H / dt_pipe / 1e6, 2),
                                    "unit": "Mpixels/s", "ms_per_step": round(dt_pipe / args.steps * 1e3, 4),
                                    "valu_issue_frac": round(6.0 * executed / (dt_pipe / args.steps) / 1e12 / peak, 4),
                                    "note": "informational: two render contexts on two streams, frames alternating; "
                                            "the headline value above runs one frame at a time"}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(w)
        else:
            # one launch = this rank's 1/N share of a frame (nu payload: 8 or 4 B/pixel; colour payload: 16);
            # duration = the library's event pair around the last launch of each render context on rank 0
            # (the lanes run concurrently, so a launch lasts longer than its share of the group time)
            ms = [m for m in lane_ms if m > 0]
            bpp = 16 if payload == "rgba" else (8 if prec == fr.Precision.F64 else 4)
            bytes_per_launch = bpp * W * fx.rows_local
            gbs = bytes_per_launch / (sum(ms) / len(ms) * 1e-3) / 1e9 if ms else 0.0
            out["exchange_verified"] = exchange_verified     # gathered frames == direct single-GPU render, on every rank
            if fr_node_obj is not None:
                out["fr_node"] = fr_node_obj
            out["roofline"] = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": None,
                               "kernel_ms": round(sum(ms) / len(ms), 4) if ms else None,
                               "note": f"per launch of a 1/{world} frame share on rank 0 ({bpp} B/pixel, {lanes} concurrent "
                                       "render contexts); the kernels are VALU-bound, see the N = 1 line for roofline_valu"}
        print(json.dumps(out), flush=True)

    if node_leg_stuck:
        # a thread of this process is stuck inside the informational leg: the line is out, leave without the destructors
        sys.stdout.flush()
        sys.stderr.write("bench.py: the fr_node leg is stuck; exiting without teardown\n")
        os._exit(0)
    if world > 1:
        for c in ctxs[1:]:
            c.close()
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
