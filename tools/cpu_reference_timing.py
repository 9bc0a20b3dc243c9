"""Time the UNMODIFIED reference (through oracle/ref_shim.py) on this host's CPU cores -- the CPU baseline SURVEY.md 8d asks
for next to the GPU number.  /root/reference only exists in the authoring container, so this script is run THERE and its
output is committed (profiles/r2_cpu_reference_container.json); bench.py quotes it beside the oracle timing it takes live on
the GPU box's host.

    python tools/cpu_reference_timing.py profiles/r2_cpu_reference_container.json
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim as R  # noqa: E402

H, F, L, HEADS = 1536, 6144, 40, 24


def main():
    out_path = sys.argv[1]
    ncores = os.cpu_count()
    torch.set_num_threads(ncores)
    torch.manual_seed(0)
    res = {"host": "authoring container", "cores": ncores, "dtype": "fp32", "torch": torch.__version__,
           "reference": "OFA-Sys/ONE-PEACE executed unmodified through oracle/ref_shim.py"}
    # (1) one 4B-dimension encoder layer, forward + backward, b = 2, every modality of the tri-modal step
    TL = R.ref("one_peace.models.transformer.transformer_layer")
    cfg = R.make_cfg(embed_dim=H, ffn_embed_dim=F, layers=1, attention_heads=HEADS, layer_scale_init_value=0.1).encoder
    layer = TL.TransformerEncoderLayer(cfg, drop_path_rate=0.0).train()
    per_layer = {}
    b = 2
    for m, S in (("text", 64), ("image", 257), ("audio", 250)):
        x = torch.randn(S, b, H, requires_grad=True)
        bias = torch.zeros(b, HEADS, S, S)
        pad = torch.zeros(b, S, dtype=torch.bool)
        ts = []
        for it in range(4):
            t0 = time.time()
            y = layer(x, pad, bias, encoder_type=m, text_seq_len=S, image_seq_len=S, audio_seq_len=S)
            y.sum().backward()
            if it:
                ts.append(time.time() - t0)
        per_layer[m] = min(ts) / b
    res["layer_fwd_bwd_s_per_sample"] = per_layer
    res["tri_modal_step_samples_per_s_extrapolated_x40_layers"] = 1.0 / (L * sum(per_layer.values()))
    res["image_text_step_samples_per_s_extrapolated_x40_layers"] = 1.0 / (L * (per_layer["text"] + per_layer["image"]))
    del layer
    # (2) the full 40-layer image tower, forward only, batch 1 (BASELINE configs[1]), measured once after a warm-up call
    RT = R.ref("one_peace.models.one_peace.one_peace_retrieval")
    mcfg = R.make_cfg(embed_dim=H, ffn_embed_dim=F, layers=L, attention_heads=HEADS, use_text_moe=False, use_audio_moe=False)
    with torch.no_grad():
        model = RT.OnePeaceRetrievalModel(mcfg, R.TinyDictionary(50265), "image").eval()
        img = torch.randn(1, 3, 256, 256)
        model(src_images=img, encoder_type="image")
        t0 = time.time()
        model(src_images=img, encoder_type="image")
        res["image_tower_40_layers_fwd_b1_s"] = time.time() - t0
        res["image_tower_images_per_s"] = 1.0 / res["image_tower_40_layers_fwd_b1_s"]
        img4 = torch.randn(1, 3, 448, 448)
        try:
            model.encoder_wrapper.image_adapter  # 448^2 needs the 28x28 bucket table: time the 256^2 tower only
        except AttributeError:
            pass
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
