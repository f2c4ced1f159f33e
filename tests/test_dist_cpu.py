"""world_size-2 (and 3) gloo tests of the N > 1 path on CPU: row-strip ownership, the gather to the
root and the root's de-interleave.  The HIP renderer cannot run here, so each rank's strips are
produced by the CPU oracle (test infrastructure standing in for the kernel); what is under test is
fractalrenderer_amd/distributed.py + the C ABI's strip arithmetic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, W, H, R, nframes, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fractalrenderer_amd.distributed import StripGather
        from oracle import oracle as O
        sg = StripGather(W, H, 4, torch.float32, torch.device("cpu"), rows_per_strip=R)

        def render_fn(shard, out, frame):
            p = O.OracleParams(max_iterations=64 + 32 * frame)
            rows = shard.global_rows(H)
            assert out.shape[0] == len(rows)
            for k, y in enumerate(rows):
                out[k] = torch.from_numpy(O.render(p, W, H, y0=int(y), y1=int(y) + 1, threads=1, planes=False).rgba[0])

        ok = True
        for f in range(nframes):
            slot = sg.submit(render_fn, f)
            sg.drain()
            if rank == 0:
                ref = O.render(O.OracleParams(max_iterations=64 + 32 * f), W, H, threads=1, planes=False).rgba
                ok = ok and np.array_equal(sg.frames[slot].numpy(), ref)
        one = sg.render_frame(render_fn, 0)
        if rank == 0:
            ref = O.render(O.OracleParams(max_iterations=64), W, H, threads=1, planes=False).rgba
            ok = ok and np.array_equal(one.numpy(), ref)
            q.put((ok, sg.R, sg.even))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,R", [(2, 40, 32, 4), (2, 24, 23, 5), (3, 16, 30, 0), (2, 8, 3, 2)])
def test_strip_gather_gloo(oracle, fr, world, W, H, R):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, R, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, used_R, even = q.get(timeout=5)
    assert ok
    if R == 0:
        assert H % (world * used_R) == 0 and even       # automatic strip height divides the frame evenly


def test_pick_rows_per_strip(fr):
    from fractalrenderer_amd.distributed import pick_rows_per_strip
    assert pick_rows_per_strip(4096, 8) == 32 and pick_rows_per_strip(8192, 8) == 32
    assert pick_rows_per_strip(4096, 1) == 32 and pick_rows_per_strip(1440, 8) == 30
    assert pick_rows_per_strip(7, 8) == 1 and pick_rows_per_strip(1000, 3) == 1


def _fx_worker(rank, world, port, W, H, R, payload, nframes, q, layout="strips"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fractalrenderer_amd.distributed import FrameExchange
        from oracle import oracle as O
        fx = FrameExchange(W, H, payload=payload, nu_dtype=torch.float64, device=torch.device("cpu"), rows_per_strip=R,
                           layout=layout)
        assert fx.bands == (layout == "bands" and H % world == 0)
        params = lambda f: O.OracleParams(max_iterations=48 + 16 * f, palette_mode=f % 6, center_x=-0.5 - 0.01 * f)  # noqa: E731

        def render_fn(shard, out, frame, plane, lane=0):
            assert plane == payload
            rows = shard.global_rows(H)
            assert out.shape[0] == len(rows)
            for k, y in enumerate(rows):
                fr_ = O.render(params(frame), W, H, y0=int(y), y1=int(y) + 1, threads=1)
                out[k] = torch.from_numpy(fr_.nu[0] if plane == "nu" else fr_.rgba[0])

        def colorize_fn(nu_frame, rgba_frame, frame):
            rgba_frame.copy_(torch.from_numpy(O.colorize(params(frame), nu_frame.numpy())))

        fx.prime()
        ok, got = True, []
        f = 0
        while f < nframes:
            count = min(world, nframes - f)
            slot = fx.submit_group(render_fn, f, count, colorize_fn if payload == "nu" else None)
            fx.drain()
            if rank < count:
                assert fx.frame_index[slot] == f + rank
                ref = O.render(params(f + rank), W, H, threads=1)
                ok = ok and np.array_equal(fx.frame_rgba[slot].numpy(), ref.rgba)
                if payload == "nu":
                    ok = ok and np.array_equal(fx.frame_nu[slot].numpy(), ref.nu)
                got.append(f + rank)
            else:
                assert fx.frame_index[slot] == -1
            f += count
        q.put((rank, ok, got))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,R,payload,nframes,layout", [
    (2, 40, 32, 4, "nu", 4, "strips"), (2, 24, 23, 5, "rgba", 3, "strips"), (3, 16, 30, 0, "nu", 7, "strips"),
    (3, 8, 2, 1, "nu", 3, "strips"), (2, 24, 23, 5, "nu", 5, "strips"),
    # rotating bands: rank r renders band (r + j) mod N of frame j, bands are received in place
    (2, 40, 32, 0, "nu", 5, "bands"), (3, 16, 30, 0, "rgba", 7, "bands"), (3, 12, 9, 0, "nu", 4, "bands"),
    (2, 24, 23, 5, "nu", 3, "bands")])                      # H % N != 0: falls back to strips
def test_frame_exchange_gloo(oracle, fr, world, W, H, R, payload, nframes, layout):
    """Rotating-root exchange: frame g*world + j must land, complete and bit-identical to a whole-frame
    render, on rank j -- even and ragged strip layouts, rotating bands, partial last group, ranks that own no rows."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fx_worker, args=(r, world, port, W, H, R, payload, nframes, q, layout)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    seen = []
    for _ in range(world):
        rank, ok, got = q.get(timeout=5)
        assert ok, rank
        seen += got
    assert sorted(seen) == list(range(nframes))          # every frame was delivered exactly once


def _bench(*flags, env_extra=None, timeout=300):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("mode,extra", [("sequence", ["--layout", "bands"]), ("sequence", ["--layout", "strips", "--rows-per-strip", "4"]),
                                        ("frame", [])])
def test_bench_launches_its_own_ranks(fr, mode, extra):
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): the process must become a launcher -- start the two
    ranks as child processes through torch.distributed.run, relay rank 0's ONE JSON line on stdout, exit 0.  Driven with
    --cpu-rehearsal (gloo, CPU tensors, stand-in planes): the launcher, the rendezvous and the exchange code are the
    real ones, no GPU is touched."""
    import json
    out = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", "--mode", mode, "--cpu-rehearsal", *extra)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 1 and d["rehearsal"] == "cpu"
    assert d["exchange_verified"] is True and d["config"]["mode"] == mode


def test_bench_launcher_hands_on_a_failing_rank(fr):
    """A rank that dies must fail the launcher (non-zero status, no JSON line), not hang it or be swallowed."""
    out = _bench("--gpus", "2", "--steps", "2", "--warmup", "0", "--cpu-rehearsal", env_extra={"FR_BENCH_REHEARSAL_FAIL_RANK": "1"})
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_bench_launcher_does_not_touch_torch_or_the_gpu_before_spawning():
    """The launcher branch must run before `import torch` (a process that has initialised the GPU may not be replaced
    or forked into ranks): statically, no torch import precedes launch_ranks() in main(), and HSA_ENABLE_IPC_MODE_LEGACY
    is set before it."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    top_imports = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    names = {a.name.split(".")[0] for n in top_imports if isinstance(n, ast.Import) for a in n.names} | \
            {(n.module or "").split(".")[0] for n in top_imports if isinstance(n, ast.ImportFrom)}
    assert "torch" not in names and "fractalrenderer_amd" not in names
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    seg = ast.get_source_segment(src, main)
    assert seg.index("HSA_ENABLE_IPC_MODE_LEGACY") < seg.index("launch_ranks(") < seg.index("import torch")


def test_bench_node_host_refuses_what_it_cannot_do(fr):
    """`bench.py --host node` is ONE process driving all devices through the C ABI: under torch.distributed.run (WORLD_SIZE set)
    it refuses instead of starting one node per rank, and without a GPU it says so (no CPU path); both before any rank is
    spawned."""
    out = _bench("--gpus", "2", "--host", "node", env_extra={"WORLD_SIZE": "2", "RANK": "0"})
    assert out.returncode != 0 and "ONE process" in (out.stderr + out.stdout)
    import torch
    if not torch.cuda.is_available():
        out = _bench("--gpus", "1", "--host", "node")
        assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)
