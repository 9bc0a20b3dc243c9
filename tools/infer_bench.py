"""Forward-only feature extraction throughput of the 4B encoder's image tower (BASELINE configs[1]: 'vision branch, image
256x256 -> 257 tokens, batch 1/8/64, forward, 1 GPU') and the long-sequence shapes of configs[4] (448^2 -> 785 tokens,
512^2 -> 1025 tokens).  Not the bench line -- a per-config throughput table for profiles/.

    python tools/infer_bench.py [--layers 40] [--iters 5] > gpurun_out/infer_bench.json
"""
import argparse
import json
import os
import sys

import torch
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (model builder + SURVEY 8d FLOP formula)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=bench.LAYERS)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--graphs", action="store_true", help="replay a captured hipGraph (one_peace_amd/graphs.py)")
    ap.add_argument("--cases", default="256:1,256:8,256:64,256:256,448:32,512:32")
    args = ap.parse_args()
    from one_peace_amd.one_peace.one_peace_retrieval import OnePeaceRetrievalModel
    from one_peace_amd.unify_model_config import one_peace_encoder_config
    dev = torch.device("cuda:0")
    rows = []
    for case in args.cases.split(","):
        res_px, b = (int(v) for v in case.split(":"))
        grid = res_px // 16
        S = grid * grid + 1
        if grid == 16:
            model = bench.build_model(args.layers, dev).eval()
        else:  # larger grid: position / relative-position buckets sized for it (the reference's ViT-style resolution change)
            enc = one_peace_encoder_config(embed_dim=bench.H, ffn_embed_dim=bench.FFN, layers=args.layers,
                                           attention_heads=bench.HEADS, drop_path_rate=0.0, layer_scale_init_value=1e-6,
                                           image_bucket_size=grid, image_rel_bucket_size=grid)
            with torch.device(dev):
                model = OnePeaceRetrievalModel(SimpleNamespace(encoder=enc, copy_rel_pos_table=False), bench._Dict(), "vl")
            model = model.to(torch.bfloat16).eval()
        imgs = torch.randn(b, 3, res_px, res_px, device=dev, dtype=torch.bfloat16)
        run = lambda: model(src_images=imgs, encoder_type="image")  # noqa: E731
        if args.graphs:
            from one_peace_amd.graphs import GraphedCall
            g = GraphedCall(lambda src_images: model(src_images=src_images, encoder_type="image"), {"src_images": imgs})
            run = lambda: g(clone=False, src_images=imgs)  # noqa: E731
        with torch.no_grad():
            for _ in range(2):
                out = run()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(args.iters):
                out = run()
            t1.record()
            torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / args.iters
        flops = b * bench.fwd_flops_per_sample(S, layers=args.layers)
        rows.append({"image": res_px, "tokens": S, "batch": b, "ms": round(ms, 3), "images_per_s": round(b / ms * 1e3, 1),
                     "tflops": round(flops / ms / 1e9, 1), "frac_of_bf16_peak": round(flops / ms / 1e9 / bench.PEAK_BF16_TFLOPS, 3),
                     "finite": bool(out.isfinite().all())})
        print(json.dumps(rows[-1]), flush=True)
        del model, imgs, out, run
        g = None
        torch.cuda.empty_cache()
    print(json.dumps({"layers": args.layers, "dtype": "bf16", "mode": "forward, no_grad, eval" + (", hipGraph replay" if args.graphs else ""), "cases": rows}))


if __name__ == "__main__":
    main()
