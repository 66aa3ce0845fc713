#!/usr/bin/env python
"""Micro-benchmark of the MSDeformAttn HIP kernels at the BASELINE call shapes.

    python tools/bench_msda.py [--iters 50] [--json out.json]

Reports, per (shape, sampling distribution, direction): average launch duration measured with HIP
events on the launch stream, algorithmic bytes and the resulting GB/s and
fraction of the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from trackformer_amd import msda  # noqa: E402

HBM_PEAK_GBS = 8000.0
CFG2_SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]


def algorithmic_bytes(N, S, M, D, L, Lq, P, backward=False, elt=4):
    val = N * S * M * D
    loc = N * Lq * M * L * P * 2
    att = N * Lq * M * L * P
    out = N * Lq * M * D
    if backward:
        return elt * (2 * val + 2 * (loc + att) + out)
    return elt * (val + loc + att + out)


def make_inputs(N, M, D, Lq, P, shapes, mode, device, seed=0, dtype=torch.float32,
                encoder_refs=False):
    g = torch.Generator().manual_seed(seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = torch.randn(N, S, M, D, generator=g, dtype=dtype)
    if encoder_refs and Lq == S:
        refs = []
        for (h, w) in shapes:
            ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5,
                                    torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
            refs.append(torch.stack([xs.reshape(-1) / w, ys.reshape(-1) / h], -1))
        ref = torch.cat(refs, 0).view(1, S, 1, 1, 1, 2).expand(N, S, 1, 1, 1, 2)
    else:
        ref = torch.rand(N, Lq, 1, 1, 1, 2, generator=g)
    if mode == "uniform":
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    elif mode == "local":       # reference point + N(0, 2 px)/size  (SURVEY.md section 8d "realistic")
        sz = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
        loc = ref + torch.randn(N, Lq, M, L, P, 2, generator=g) * 2.0 / sz
    elif mode == "init":        # what MSDeformAttn._reset_parameters produces (zero weight, grid bias)
        dirs = torch.tensor([(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1) if (a, b) != (0, 0)],
                            dtype=torch.float32).view(1, 1, M, 1, 1, 2)
        k = torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, 1, 1, P, 1)
        hw = torch.tensor([[h, w] for h, w in shapes], dtype=torch.float32).view(1, 1, 1, L, 1, 2)
        loc = ref + dirs * k / hw   # (H, W) divisor quirk of ms_deform_attn.py:79
        loc = loc.expand(N, Lq, M, L, P, 2).contiguous()
    else:
        raise ValueError(mode)
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    grad_out = torch.randn(N, Lq, M * D, generator=g)
    shapes_t = torch.tensor(shapes, dtype=torch.long, device=device)
    msda.attach_host_shapes(shapes_t, shapes)
    return (value.to(device), shapes_t, loc.to(dtype).to(device), attn.to(dtype).to(device),
            grad_out.to(dtype).to(device))


def time_launches(fn, iters, warmup=5):
    """Average duration (ms) of one launch: `iters` launches captured in ONE HIP graph (so that host
    launch overhead cannot hide in the number), HIP events around the graph replay."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            for _ in range(iters):
                fn()
        graph.replay()
        stream.synchronize()
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record(stream)
        graph.replay()
        end.record(stream)
        end.synchronize()
    return start.elapsed_time(end) / iters


CONFIGS = [
    # name, dict(N, M, D, Lq, P, shapes)
    ("cfg2_encoder", dict(N=1, M=8, D=32, Lq=22223, P=4, shapes=CFG2_SHAPES)),
    ("cfg2_decoder", dict(N=1, M=8, D=32, Lq=400, P=4, shapes=CFG2_SHAPES)),
    ("cfg4_decoder", dict(N=1, M=8, D=36, Lq=800, P=4, shapes=CFG2_SHAPES * 2)),
    ("cfg3_encoder_n2", dict(N=2, M=8, D=32, Lq=22223, P=4, shapes=CFG2_SHAPES)),
]


def run(iters=50, modes=("uniform", "local", "init"), device="cuda:0", backward=True, only=None, forward=True):
    rows = []
    for name, kw in CONFIGS:
        if only and name not in only:
            continue
        for mode in modes:
            S = sum(h * w for h, w in kw["shapes"])
            value, shapes, loc, attn, grad_out = make_inputs(
                mode=mode, device=device, encoder_refs=(kw["Lq"] == S), **kw)
            dims = dict(N=kw["N"], S=S, M=kw["M"], D=kw["D"], L=len(kw["shapes"]), Lq=kw["Lq"],
                        P=kw["P"])
            if forward:
                ms = time_launches(lambda: msda.ms_deform_attn_forward(value, shapes, loc, attn, 64),
                                   iters)
                b = algorithmic_bytes(**dims)
                rows.append(dict(shape=name, mode=mode, dir="fwd", ms=ms, alg_MB=b / 1e6,
                                 GBs=b / ms / 1e6, frac=b / ms / 1e6 / HBM_PEAK_GBS))
            if backward:
                ms = time_launches(lambda: msda.ms_deform_attn_backward(value, shapes, loc, attn,
                                                                        grad_out, 64), iters)
                b = algorithmic_bytes(backward=True, **dims)
                rows.append(dict(shape=name, mode=mode, dir="bwd", ms=ms, alg_MB=b / 1e6,
                                 GBs=b / ms / 1e6, frac=b / ms / 1e6 / HBM_PEAK_GBS))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-backward", action="store_true")
    ap.add_argument("--no-forward", action="store_true")
    ap.add_argument("--shapes", default=None, help="comma-separated subset of: " + ", ".join(n for n, _ in CONFIGS))
    ap.add_argument("--modes", default="uniform,local,init")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="tf_msda_set_option before timing, e.g. --option direct9=1 --option pquad=0")
    args = ap.parse_args()
    from trackformer_amd import _cabi
    for o in args.option:
        k, v = o.split("=")
        print("option %s: %d -> %s" % (k, _cabi.lib().tf_msda_set_option(k.encode(), int(v)), v))
    rows = run(args.iters, modes=tuple(args.modes.split(",")), backward=not args.no_backward,
               only=args.shapes.split(",") if args.shapes else None, forward=not args.no_forward)
    for r in rows:
        print("%-16s %-8s %s  %8.1f us  %7.2f MB  %8.1f GB/s  %5.1f%% of HBM peak" % (
            r["shape"], r["mode"], r["dir"], r["ms"] * 1e3, r["alg_MB"], r["GBs"],
            100 * r["frac"]))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
