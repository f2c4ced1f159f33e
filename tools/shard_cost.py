#!/usr/bin/env python3
"""What one rank of an N-GPU run does per group of N frames: N renders of its 1/N row-strip share
(tools for DESIGN.md section 6; single GPU).  Prints ms per group for plane nu / rgba, sequential on one
stream and spread over several contexts+streams, and the strong-scaling bound it implies."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--max-iter", type=int, default=1024)
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to fr_ctx_set_option on every context")
    ap.add_argument("--only", type=int, default=0, help="only this N")
    ap.add_argument("--bands", action="store_true", help="rotating bands: the N renders of a group are the N bands of H/N rows")
    args = ap.parse_args()
    import torch
    import fractalrenderer_amd as fr
    from fractalrenderer_amd.distributed import pick_rows_per_strip
    dev = torch.device("cuda:0")
    W = H = args.size
    st = fr.FractalState(max_iterations=args.max_iter)
    nctx = 8
    rs = [fr.Renderer(0) for _ in range(nctx)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nctx)]
    for o in args.opt:
        k, v = o.split("=")
        for r_ in rs:
            r_.set_option(k, int(v))

    cur = torch.cuda.Stream(device=dev)     # a real stream: handle 0 would mean "the context's own stream"

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(args.reps):
            fn()
        e1.record(cur)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    full = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
    base = timed(lambda: rs[0].render(st, W, H, rgba=full, sync=False, stream=cur.cuda_stream))
    print("1 GPU whole frame rgba: %.3f ms" % base)
    for N in ((args.only,) if args.only else (1, 2, 4, 8)):
        R = H // N if args.bands else pick_rows_per_strip(H, N)
        sh = fr.Shard(0, N, R)
        rows = sh.rows(H)
        shard_of = (lambda j: fr.Shard(j, N, R)) if args.bands else (lambda j: sh)
        for plane in ("nu", "rgba"):
            bufs = [torch.empty((rows, W) + ((4,) if plane == "rgba" else ()),
                                dtype=torch.float32 if plane == "rgba" else torch.float64, device=dev) for _ in range(N)]

            def seq():
                for j in range(N):
                    rs[0].render(st, W, H, shard=shard_of(j), sync=False, stream=cur.cuda_stream, **{plane: bufs[j]})

            def multi(k):
                def fn():
                    evs = []
                    for j in range(N):
                        s = streams[j % k]
                        if j < k:
                            s.wait_stream(cur)
                        rs[j % k].render(st, W, H, shard=shard_of(j), sync=False, stream=s.cuda_stream, **{plane: bufs[j]})
                    for s in streams[:k]:
                        cur.wait_stream(s)
                return fn

            t1 = timed(seq)
            line = "N=%d plane=%-4s  seq %.3f ms/group (%.2fx)" % (N, plane, t1, N * base / t1)
            for k in (2, 4, 8):
                if N >= k:
                    tk = timed(multi(k))
                    line += "   %d streams %.3f ms (%.2fx)" % (k, tk, N * base / tk)
            print(line, flush=True)
    # destination side of a group: de-interleave one frame's nu strips and recolour it
    nu = torch.empty((H, W), dtype=torch.float64, device=dev)
    rgba = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
    rs[0].render(st, W, H, nu=nu)
    with torch.cuda.stream(cur):
        t = timed(lambda: rs[0].colorize(st, nu, rgba, stream=cur.cuda_stream))
        print("colorize %dx%d: %.3f ms (%.0f GB/s of 24 B/pixel)" % (W, H, t, 24 * W * H / t / 1e6))
        for N in (2, 4, 8):
            R = pick_rows_per_strip(H, N)
            S = H // (N * R)
            parts = [torch.empty((S * R, W), dtype=torch.float64, device=dev) for _ in range(N)]
            fv = nu.view(S, N, R, W)

            def asm():
                for p_, part in enumerate(parts):
                    fv[:, p_].copy_(part.view(S, R, W))
            t = timed(asm)
            print("assemble N=%d: %.3f ms (%.0f GB/s of 16 B/pixel)" % (N, t, 16 * W * H / t / 1e6))
    for r in rs:
        r.close()


if __name__ == "__main__":
    main()
