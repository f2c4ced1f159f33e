#!/usr/bin/env python3
"""The C5 pipeline end to end on one card through the C ABI: the reference's .franim, 8192x8192, max_iter 4096 override, every
100th of its 2400 frames (24 frames), fr_node_render_animation (render over the node's parts -> 8-bit export on the root ->
3 B/pixel back -> PNG on a writer thread), against the same frames one at a time through fr_render_frame_png.
usage: anim_sweep.py [size] [parts]"""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fractalrenderer_amd as fr
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
anim = fr.AnimationSystem()
assert anim.load_from_file(os.path.join(ROOT, "tests", "golden", "reference_sample.franim"))
frames = list(range(0, anim.frame_count(), 100))
tmp = tempfile.mkdtemp(prefix="fr_anim_")
try:
    with fr.Renderer(0) as r:
        t0 = time.perf_counter()
        for f in frames[:6]:
            st = anim.interpolate(anim.frame_time(f)); st.max_iterations = 4096
            assert r.render_frame(st, size, size, os.path.join(tmp, "one_%06d.png" % f), precision=fr.Precision.F64)
        one = (time.perf_counter() - t0) / 6
    print(f"{size}x{size} max_iter 4096 fp64: fr_render_frame_png one frame at a time {one * 1e3:.1f} ms/frame (render + export + copy back + PNG, serial)")
    for slots in (1, 2, 4):
        with fr.Node([0] * parts) as node:
            node.set_option("slots", slots)
            out = os.path.join(tmp, f"node{slots}")
            t0 = time.perf_counter()
            n = node.render_animation(anim, out, width=size, height=size, frame_step=100, precision=fr.Precision.F64, max_iterations=4096)
            dt = time.perf_counter() - t0
        same = open(os.path.join(out, "frame_000500.png"), "rb").read() == open(os.path.join(tmp, "one_000500.png"), "rb").read()
        mb = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) / n / 1e6
        print(f"  fr_node_render_animation, {parts} part(s) on device 0, {slots} slot(s): {n} frames in {dt:.2f} s = {dt / n * 1e3:.1f} ms/frame "
              f"({size * size * n / dt / 1e6:.0f} Mpx/s end to end, PNG files included, {mb:.1f} MB per file, FR_PNG_LEVEL={os.environ.get('FR_PNG_LEVEL', '6 (default)')}); frame 500 identical: {same}")
        shutil.rmtree(out)
    # the same frames as packed RGB24 into a file descriptor (fr_anim_render_options.raw_fd: an encoder's stdin) -- here a pipe
    # whose reader throws the bytes away, and /dev/null: what the devices + the 3 B/pixel copy back deliver without a deflate
    import threading
    for slots in (2, 4):
        for sink in ("pipe", "devnull"):
            if sink == "pipe":
                rd, wr = os.pipe()
                def drain():
                    with os.fdopen(rd, "rb", buffering=0) as f:
                        while f.read(1 << 22): pass
                th = threading.Thread(target=drain); th.start()
            else:
                wr = os.open(os.devnull, os.O_WRONLY); th = None
            with fr.Node([0] * parts) as node:
                node.set_option("slots", slots)
                t0 = time.perf_counter()
                n = node.render_animation(anim, None, width=size, height=size, frame_step=100, precision=fr.Precision.F64, max_iterations=4096, raw_fd=wr)
                dt = time.perf_counter() - t0
            os.close(wr)
            if th: th.join()
            print(f"  fr_node_render_animation, raw RGB24 into {'a pipe (reader discards)' if sink == 'pipe' else '/dev/null'}, {parts} part(s), {slots} slot(s): "
                  f"{n} frames in {dt:.2f} s = {dt / n * 1e3:.1f} ms/frame ({size * size * n / dt / 1e6:.0f} Mpx/s end to end)")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
