"""Benchmark of the hot path on MI355X: one JSON line per run (rank 0), one mode per BASELINE.json config.

    python bench.py --gpus 1 --steps 10 --warmup 2                     # headline: configs[3], the tri-modal pretrain step
    python bench.py --config 1 [--batch 64]                            # configs[1]: vision-branch image-only forward, 256^2
    python bench.py --config 2                                         # configs[2]: image+text contrastive step, b=256/GPU
    python bench.py --config 4 [--res 448|512] [--fp8]                 # configs[4]: long-sequence image (+text) step
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" of configs 2-4 is the full training step on one synthetic batch that is already resident in HBM: the
single-modality forwards through the 40-layer H=1536 encoder with the per-modality FFN sets, one fused all-gather of the
[k, b, H] embeddings, the contrastive losses, backward, bucketed gradient all-reduce (overlapped with backward), global-norm
clipping and the fused AdamW update.  A step of config 1 is one no-grad forward of the image tower.  Per-GPU batch is fixed
(weak scaling).  (configs[0], the tiny text model on the CPU reference path, is a parity case: tests/test_model_cpu.py.)
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

# the host driver of the MI355X pool supports dmabuf IPC only: without this RCCL's buffer exchange between the ranks of a node fails with
# "hipIpcGetMemHandle: invalid argument".  Exported by the image already; set here too, before the HIP runtime comes up.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, FFN, LAYERS, HEADS = 1536, 6144, 40, 24
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 MFMA (MX-scaled K=128 instructions), same guide
PROFILE_EVERY = 10         # HIP-event pairs around every GEMM / attention launch on every 10th timed step: 1 700 events cost a step
                           # +24 ms (measured, with or without the system-scope fence: each record is a barrier packet of its own)


def fwd_flops_per_sample(S, layers=LAYERS, h=H, f=FFN):
    """per token per layer 8H^2 + 4SH + 6HF; per sample L*S*that."""
    return layers * S * (8 * h * h + 4 * S * h + 6 * h * f)


def audio_adapter_fwd_flops(seconds):
    return 31.5e9 * seconds / 5.0  # SURVEY.md 8d (31.5 GFLOP @ 5 s)


def audio_frames(n):
    for k, s in [(10, 5)] + [(3, 2)] * 4 + [(2, 2)] * 2:
        n = (n - k) // s + 1
    return n


class _Dict:
    def __len__(self):
        return 50265

    def pad(self):
        return 1


def build_model(layers, device, recompute=False, head="val", image_grid=16, extra=None):
    """ONE-PEACE-4B encoder as the retrieval model (contrastive heads) with the FFN sets of `head` ('val': text + image +
    audio, 'vl': text + image, 'image': image only); image_grid = patches per side (16: 256^2, 28: 448^2, 32: 512^2)."""
    from one_peace_amd.one_peace.one_peace_retrieval import OnePeaceRetrievalModel
    from one_peace_amd.unify_model_config import one_peace_encoder_config
    kw = dict(embed_dim=H, ffn_embed_dim=FFN, layers=layers, attention_heads=HEADS, drop_path_rate=0.4,
              layer_scale_init_value=1e-6, audio_bucket_size=512, checkpoint_activations=recompute)
    if image_grid != 16:  # larger grid: position / relative-position buckets sized for it (ViT-style resolution change)
        kw.update(image_bucket_size=image_grid, image_rel_bucket_size=image_grid)
    kw.update(extra or {})
    enc = one_peace_encoder_config(**kw)
    cfg = SimpleNamespace(encoder=enc, copy_rel_pos_table=False)
    with torch.device(device):
        model = OnePeaceRetrievalModel(cfg, _Dict(), head)
    return model.to(torch.bfloat16)


def build_pretrain_vl_model(layers, device, recompute=False):
    """pretrain_vl_3B.yaml: the 4B encoder (text + image towers) plus the 2-layer 768-wide decoder of the masked branch."""
    from one_peace_amd.one_peace.one_peace_pretrain import OnePeacePretrainModel
    from one_peace_amd.unify_model_config import one_peace_encoder_config
    enc = one_peace_encoder_config(embed_dim=H, ffn_embed_dim=FFN, layers=layers, attention_heads=HEADS, drop_path_rate=0.4,
                                   layer_scale_init_value=1e-6, use_audio_moe=False, checkpoint_activations=recompute)
    dec = one_peace_encoder_config(embed_dim=768, ffn_embed_dim=2048, layers=2, attention_heads=12, drop_path_rate=0.0,
                                   use_audio_moe=False, checkpoint_activations=recompute)
    dec.text_adapter.use_attn_bias = dec.image_adapter.use_attn_bias = False
    dec.image_adapter.vision_encoder_type = "none"
    dec.use_layer_scale = False
    cfg = SimpleNamespace(encoder=enc, decoder=dec, copy_rel_pos_table=False, reset_logit_scale=False,
                          logit_scale_init=1 / 0.07, stage2_pretrain=False)
    with torch.device(device):
        model = OnePeacePretrainModel(cfg, _Dict())
    return model.to(torch.bfloat16).train()


def build_pretrain_al_model(layers, device, recompute=False, stage2=True):
    """pretrain_al_3B.yaml: the 4B encoder with text + audio towers, the 2-layer 768-wide decoder with the spec-less fixed-position
    audio adapter and no layer scale, `stage2_pretrain: true` (one_peace_pretrain.py:98-104: only the audio adapter, the audio
    FFNs and audio_layer_norm of the encoder train; text_proj frozen)."""
    from one_peace_amd.one_peace.one_peace_pretrain import OnePeacePretrainModel
    from one_peace_amd.unify_model_config import one_peace_encoder_config
    enc = one_peace_encoder_config(embed_dim=H, ffn_embed_dim=FFN, layers=layers, attention_heads=HEADS, drop_path_rate=0.4,
                                   layer_scale_init_value=1e-6, use_image_moe=False, audio_bucket_size=512,
                                   checkpoint_activations=recompute)
    dec = one_peace_encoder_config(embed_dim=768, ffn_embed_dim=2048, layers=2, attention_heads=12, drop_path_rate=0.0,
                                   use_image_moe=False, checkpoint_activations=recompute)
    dec.text_adapter.use_attn_bias = dec.audio_adapter.use_attn_bias = False
    dec.audio_adapter.feature_encoder_spec = None
    dec.audio_adapter.abs_pos_type = "fixed"
    dec.audio_adapter.bucket_size = 256
    dec.use_layer_scale = False
    cfg = SimpleNamespace(encoder=enc, decoder=dec, copy_rel_pos_table=False, reset_logit_scale=True,
                          logit_scale_init=1 / 0.07, stage2_pretrain=stage2)
    with torch.device(device):
        model = OnePeacePretrainModel(cfg, _Dict())
    return model.to(torch.bfloat16).train()


def add_pretrain_al_masks(batch, seed):
    """Preserve ids / mask indices of the audio-language stage with the mask ratios of pretrain_al_3B.yaml:12-14 (position 0 = CLS
    is always kept; synthetic clips have no padding)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    tok = batch["src_tokens"].cpu()
    b = tok.shape[0]
    dev = batch["src_tokens"].device

    def make(valid, ratio):
        S = valid.shape[1]
        rows, mask = [], torch.zeros(b, S, dtype=torch.bool)
        for i in range(b):
            cand = torch.nonzero(valid[i, 1:]).flatten() + 1
            n_mask = int(len(cand) * ratio)
            perm = cand[torch.randperm(len(cand), generator=g)]
            mask[i, perm[:n_mask]] = True
            rows.append(torch.cat([torch.zeros(1, dtype=torch.long), perm[n_mask:].sort().values]))
        K = max(len(r) for r in rows)
        ids = torch.full((b, K), -1, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
        return ids.to(dev), mask.to(dev)
    text_valid = torch.cat([torch.ones(b, 1, dtype=torch.bool), tok.ne(1)], dim=1)
    audio_valid = ~batch["audio_padding_masks"].cpu()
    out = dict(batch)
    out["audio_preserve_ids"], out["audio_mask_indices"] = make(audio_valid, 0.55)
    out["al_text_preserve_ids"], out["al_text_mask_indices"] = make(text_valid, 0.4)
    out["al_audio_preserve_ids"], out["al_audio_mask_indices"] = make(audio_valid, 0.45)
    out.pop("src_images", None)
    return out


def add_pretrain_masks(batch, seed):
    """Preserve ids / mask indices in the form of data/pretrain_data/image_text_pretrain_dataset.py:85-117 with the mask
    ratios of pretrain_vl_3B.yaml:13-16 (position 0 = CLS is always kept)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    tok = batch["src_tokens"].cpu()
    b = tok.shape[0]
    dev = batch["src_tokens"].device

    def make(valid, ratio):
        S = valid.shape[1]
        rows, mask = [], torch.zeros(b, S, dtype=torch.bool)
        for i in range(b):
            cand = torch.nonzero(valid[i, 1:]).flatten() + 1
            n_mask = int(len(cand) * ratio)
            perm = cand[torch.randperm(len(cand), generator=g)]
            mask[i, perm[:n_mask]] = True
            rows.append(torch.cat([torch.zeros(1, dtype=torch.long), perm[n_mask:].sort().values]))
        K = max(len(r) for r in rows)
        ids = torch.full((b, K), -1, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
        return ids.to(dev), mask.to(dev)
    text_valid = torch.cat([torch.ones(b, 1, dtype=torch.bool), tok.ne(1)], dim=1)
    image_valid = torch.ones(b, 257, dtype=torch.bool)
    out = dict(batch)
    out["text_preserve_ids"], out["text_mask_indices"] = make(text_valid, 0.15)
    out["image_preserve_ids"], out["image_mask_indices"] = make(image_valid, 0.75)
    out["vl_text_preserve_ids"], out["vl_text_mask_indices"] = make(text_valid, 0.4)
    out["vl_image_preserve_ids"], out["vl_image_mask_indices"] = make(image_valid, 0.6875)
    return out


def synthetic_batch(b, device, seed, res=256, audio_seconds=None, text=True, text_len=63, dtype=torch.bfloat16):
    """SURVEY.md 8d synthetic inputs: tokens uniform in [4, 50264] with r % 8 trailing pads, N(0,1) pixels / waveforms."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    if text:
        tok = torch.randint(4, 50265, (b, text_len), generator=g)
        for i in range(b):
            k = i % 8
            if k:
                tok[i, text_len - k:] = 1
        out["src_tokens"] = tok.to(device)
    out["src_images"] = torch.randn(b, 3, res, res, generator=g).to(device).to(dtype)
    audio_S = 0
    if audio_seconds:
        n_wav = int(16000 * audio_seconds)
        audio_S = audio_frames(n_wav) + 1
        out["src_audios"] = torch.randn(b, n_wav, generator=g).to(device).to(dtype)
        out["audio_padding_masks"] = torch.zeros(b, audio_S, dtype=torch.bool, device=device)
    return out, audio_S


def _container_reference():
    """The reference itself (unmodified, through oracle/ref_shim.py) timed in the authoring container, where /root/reference
    exists: tools/cpu_reference_timing.py wrote the file; it is quoted here, never measured on the GPU box."""
    p = os.path.join(ROOT, "profiles", "r2_cpu_reference_container.json")
    return json.load(open(p)) if os.path.exists(p) else None


def cpu_baseline(modalities, backward=True, seconds_budget=20.0):
    """The oracle (CPU restatement of the reference, fp32, host cores of THIS machine) on a bounded sample of the same
    workload: one 4B-dimension encoder layer (forward + backward, or forward only) for each modality's sequence length at
    b = 2, timed, and extrapolated x40 layers to samples/s (adapters and the contrastive head are < 5 % and left out).
    `modalities`: {name: S}."""
    from oracle import onepeace_oracle as O
    torch.manual_seed(0)
    ncores = min(os.cpu_count() or 1, 32)  # more threads than this only adds fork/join overhead at these sizes
    torch.set_num_threads(ncores)
    p = "L"
    sd = {}

    def mk(name, *shape, scale=0.02):
        sd[p + "." + name] = (torch.randn(*shape) * scale).requires_grad_(True)
    for n in ("self_attn_layer_norm", "final_layer_norm", "self_attn.ln"):
        sd[p + "." + n + ".weight"] = torch.ones(H, requires_grad=True)
        sd[p + "." + n + ".bias"] = torch.zeros(H, requires_grad=True)
    for n in ("q_proj", "v_proj", "out_proj"):
        mk("self_attn.%s.weight" % n, H, H)
        mk("self_attn.%s.bias" % n, H)
    mk("self_attn.k_proj.weight", H, H)
    sd[p + ".gamma_1"] = torch.full((H,), 0.1, requires_grad=True)
    sd[p + ".gamma_2"] = torch.full((H,), 0.1, requires_grad=True)
    for m in modalities:
        mk(m + "_ffn.0.wi_0.weight", FFN, H)
        mk(m + "_ffn.0.wi_1.weight", FFN, H)
        sd[p + "." + m + "_ffn.2.weight"] = torch.ones(FFN, requires_grad=True)
        sd[p + "." + m + "_ffn.2.bias"] = torch.zeros(FFN, requires_grad=True)
        mk(m + "_ffn.3.weight", H, FFN)
        mk(m + "_ffn.3.bias", H)
    b = 2
    per_sample = 0.0
    t_start = time.time()
    detail = {}
    for m, S in modalities.items():
        x = torch.randn(S, b, H, requires_grad=backward)
        bias = torch.zeros(b, HEADS, S, S)
        times = []
        for it in range(4):  # first pass = warm-up (allocator, thread pool), not timed
            t0 = time.time()
            if backward:
                O.encoder_layer(x, sd, p, HEADS, m, bias).sum().backward()
            else:
                with torch.no_grad():
                    O.encoder_layer(x, sd, p, HEADS, m, bias)
            if it > 0:
                times.append(time.time() - t0)
            if it > 0 and time.time() - t_start > seconds_budget:
                break
        detail[m] = min(times) / b
        per_sample += detail[m]
    out = {"value": 1.0 / (LAYERS * per_sample), "unit": "samples/s", "cores": ncores, "kind": "port",
           "sample": "oracle (fp32 torch-CPU restatement of the reference) 1 encoder layer %s at H=1536/F=6144, b=2, %s, best of "
                     "<=3 after a warm-up pass; EXTRAPOLATED x40 layers (per-layer s/sample: %s)" % (
                         "fwd+bwd" if backward else "fwd", " + ".join("%s S=%d" % kv for kv in modalities.items()),
                         json.dumps({k: round(v, 4) for k, v in detail.items()}))}
    ref = _container_reference()
    if ref is not None:
        out["reference_in_authoring_container"] = ref  # the unmodified reference, measured where /root/reference exists
    return out


def auto_batch(device, tokens_per_sample, layers, recompute, candidates, fixed_gb=50.0, reserve_gb=0.0):
    """Largest candidate whose kept activations fit: 46.6 GB of parameters / gradients / Adam moments (4B model) + 67.6 KB per
    token per layer measured on MI355X (0.25 GB per 571-token tuple under recompute); `reserve_gb` stays free for RCCL's
    transport buffers (device memory outside torch's allocator).  Same choice on every rank."""
    total_gb = torch.cuda.get_device_properties(device).total_memory / 1e9
    per_tok_gb = (0.25 / 571 if recompute else 1.64 / 571) * 1.0737 * layers / LAYERS
    for b in candidates:
        if fixed_gb + per_tok_gb * tokens_per_sample * b <= 0.93 * total_gb - reserve_gb:
            return b
    return candidates[-1]


def param_checksum(flat):
    """[sum, sum of absolute values] of the flat parameter buffer in fp64, 128 M elements at a time (a whole-buffer fp64 reduction made
    torch allocate a 31 GB temporary: +29 GB peak reserved in the bench line)."""
    tot = mag = 0.0
    for c in flat.split(1 << 27):
        tot += float(c.sum(dtype=torch.float64))
        mag += float(c.abs().sum(dtype=torch.float64))
    return [tot, mag]


def main():
    global H, FFN, LAYERS, HEADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 4],
                    help="index into BASELINE.json configs: 1 image-only forward, 2 image+text contrastive step (b=256/GPU), "
                         "3 tri-modal pretrain step (headline), 4 long-sequence image(+text) step at 448^2 / 512^2")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch; 0 = the config's own (largest that fits this GPU's HBM)")
    ap.add_argument("--res", type=int, default=448, choices=[448, 512], help="config 4: image size (785 / 1025 tokens)")
    ap.add_argument("--fp8", action="store_true", help="config 4: FFN GEMMs on the fp8 (e4m3) MFMA path: forward up- / down-projection and "
                    "(round 6) the two input-gradient GEMMs of the backward; weight gradients stay bf16")
    ap.add_argument("--fp8-forward-only", action="store_true", help="with --fp8: keep the FFN input-gradient GEMMs in bf16 (round 5's variant; A/B)")
    ap.add_argument("--layers", type=int, default=LAYERS, help="debug only; the reported metric needs 40")
    ap.add_argument("--audio-seconds", type=float, default=5.0)
    ap.add_argument("--recompute", action="store_true", help="per-layer activation recompute (reference default)")
    ap.add_argument("--recompute-cheap", action="store_true",
                    help="keep the layer activations EXCEPT the four LayerNorm-type outputs only weight gradients read; backward re-creates "
                         "them (ops.set_recompute_cheap: -63 GB at the headline batch for +3 LayerNorm / +1 LN-GeGLU passes per layer).  "
                         "Switched on automatically, before the per-GPU batch is halved, when the free memory after model + optimiser + "
                         "RCCL set-up does not hold every activation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loss-curve", action="store_true", help="record the loss of every timed step in config.loss_curve (same batch every step)")
    ap.add_argument("--no-lock-step", action="store_true",
                    help="contrastive configs: one encoder pass per modality (the reference's call pattern) instead of the lock-step pass")
    ap.add_argument("--skip-dropped", action="store_true",
                    help="stochastic depth on the kept samples only (TransformerEncoder.skip_dropped_branches): the timed step does not "
                         "compute the branch outputs the reference multiplies by zero; default: the reference's arithmetic")
    ap.add_argument("--no-skip-leg", action="store_true",
                    help="single-GPU headline config: do not time the extra leg with skip_dropped_branches that is reported beside `value`")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the 1 s register-only MFMA loop behind roofline.power_limited_peak")
    ap.add_argument("--objective", choices=["contrastive", "pretrain-vl", "pretrain-al"], default="contrastive",
                    help="config 3 only: contrastive = the headline tri-modal ITC+ATC step; pretrain-vl = the full image-text "
                         "pretraining objective (ITC + four DCL terms, six passes incl. the masked students and the decoder); "
                         "pretrain-al = the stage-2 audio-language objective (ATC + three DCL terms; everything but the audio "
                         "adapter / audio FFNs / audio_layer_norm of the encoder frozen, pretrain_al_3B.yaml)")
    ap.add_argument("--check-replicas", action="store_true",
                    help="after the run, compare a checksum of all parameters across ranks (every rank sees different data, so the "
                         "replicas only stay identical if every gradient was all-reduced after its last contribution)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="every step takes its batch from host memory through staging.SamplePrefetcher (PCIe-inclusive rate; "
                         "the headline value keeps inputs resident in HBM)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--debug-cpu-micro", action="store_true",
                    help="NOT a measurement: the same control flow (init, broadcast, bucketed reducer, fused all-gather criterion, "
                         "replica / launch-order checks, max-over-ranks timing, JSON line) on CPU over gloo with a 2-layer H=128 model, "
                         "the torch path of the mirrors and optim.TorchAdamW -- what the world-size-4 test of tests/ runs")
    ap.add_argument("--graphs", action="store_true",
                    help="single process, training configs: zero-grad + forwards + loss + backward of a step replayed as ONE "
                         "hipGraph (graphs.TrainStepGraph; the optimiser step stays eager).  For launch-bound small batches; the "
                         "per-launch HIP-event roofline needs eager launches and is omitted")
    args = ap.parse_args()

    from one_peace_amd import hip
    from one_peace_amd.criterions.contrastive import ImageTextRetrievalCriterion, TriModalContrastiveCriterion
    from one_peace_amd.distributed import BucketedGradReducer, FlatParameters, init_distributed
    from one_peace_amd.optim import FusedAdamW, TorchAdamW

    micro = args.debug_cpu_micro
    if micro:
        H, FFN, LAYERS, HEADS = 128, 256, 2, 2
        args.layers, args.no_profile, args.no_cpu_baseline, args.audio_seconds = LAYERS, True, True, 0.5
        args.batch = args.batch or 4
        if args.config != 3 or args.objective != "contrastive" or args.graphs or args.host_inputs or args.fp8:
            raise SystemExit("--debug-cpu-micro runs the headline control flow only")
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    else:
        hip.lib()
    rank, world, local = init_distributed("gloo" if micro else os.environ.get("ONEPEACE_DIST_BACKEND"))
    assert world == args.gpus, "launch with --nproc-per-node == --gpus"
    if os.environ.get("ONEPEACE_SINGLE_DEVICE_DEBUG"):  # functional test of the N>1 control flow on a 1-GPU box (gloo)
        local = 0
    device = torch.device("cpu") if micro else torch.device("cuda", local)
    if not micro:
        torch.cuda.set_device(device)
    torch.manual_seed(3407 + rank)
    full = args.config == 3 and args.objective in ("pretrain-vl", "pretrain-al")
    al = args.config == 3 and args.objective == "pretrain-al"
    train = args.config != 1
    rccl_reserve = 6.0 if world > 1 else 0.0  # GB kept free for RCCL transport buffers / rings on a multi-GPU node

    # ---------------------------------------------------------------- workload of the selected BASELINE config
    S_txt = 64
    if args.config == 1:
        S_img, audio_s, head, res = 257, None, "image", 256
        args.batch = args.batch or 64
        modal = {"image": S_img}
        name = "BASELINE configs[1]: vision branch of ONE-PEACE-4B (H=1536, L=40, image FFNs), image-only forward, 256^2 -> 257 tokens"
        metric = "image-tower forward images/s ONE-PEACE-4B vision branch"
    elif args.config == 2:
        S_img, audio_s, head, res = 257, None, "vl", 256
        if args.batch <= 0:
            args.batch = 256  # the batch configs[2] names; its kept activations (236 GB) + model state do not fit 288 GB,
            total_gb = torch.cuda.get_device_properties(device).total_memory / 1e9
            kept_all = (1.64 / 571) * 1.0737 * (S_img + S_txt) * 256 * args.layers / LAYERS
            if 36.0 + kept_all > 0.93 * total_gb - rccl_reserve and not args.recompute_cheap:
                # (round 5) ... but they do at the cheap level (32 / 46 of the bytes: the LayerNorm-type outputs are re-created in backward,
                # +~5 % step time); only when not even that fits do the layers recompute entirely (the reference's
                # checkpoint_activations: true, a whole extra forward per step)
                if 36.0 + kept_all * 32.0 / 46.0 <= 0.93 * total_gb - rccl_reserve - 20.0 and not args.recompute and not args.graphs:
                    args.recompute_cheap = True
                else:
                    args.recompute = True
        modal = {"text": S_txt, "image": S_img}
        name = ("BASELINE configs[2]: ONE-PEACE-4B image+text contrastive step (image_text_retrieval_criterion: 2 forwards, fused "
                "[2,b,H] all-gather, ITC), backward, grad all-reduce, grad-norm clip, AdamW")
        metric = "image-text contrastive step samples/s ONE-PEACE-4B"
    elif args.config == 4:
        res = args.res
        grid = res // 16
        S_img, audio_s, head = grid * grid + 1, None, "vl"
        if args.batch <= 0:
            args.batch = auto_batch(device, S_img + S_txt, args.layers, args.recompute, (64, 48, 32, 16, 8), 36.0, rccl_reserve)
        modal = {"text": S_txt, "image": S_img}
        name = ("BASELINE configs[4]: ONE-PEACE-4B long-sequence step, %d^2 image -> %d tokens + text 64: 2 forwards, ITC, backward, "
                "grad all-reduce, grad-norm clip, AdamW%s" % (res, S_img, ("; FFN GEMMs in fp8 (e4m3, per-row scales): forward" + ("" if args.fp8_forward_only else " + input gradients")) if args.fp8 else ""))
        metric = "long-sequence (%d-token image) contrastive step samples/s ONE-PEACE-4B" % S_img
    else:
        S_img, audio_s, head, res = (17 if micro else 257), (None if full and not al else args.audio_seconds), "val", (64 if micro else 256)
        if micro:
            S_txt = 16
        audio_S = 0 if full and not al else audio_frames(int(16000 * audio_s)) + 1
        if args.batch <= 0:
            args.batch = auto_batch(device, S_img + S_txt + audio_S, args.layers, args.recompute, (128, 64, 32, 16, 8), 50.0,
                                    rccl_reserve)
        if full and args.batch > 64:
            args.batch = 64  # five passes keep activations (two teachers, three students): 64 tuples fit
        modal = ({"text": S_txt, "audio": audio_S} if al else {"text": S_txt, "image": S_img} if full else
                 {"text": S_txt, "image": S_img, "audio": audio_S})
        name = ("BASELINE configs[3]: ONE-PEACE-4B tri-modal (image 256^2 + text 64 + audio %.0fs) contrastive pretrain step: 3 forwards, "
                "ITC+ATC, backward, grad all-reduce, grad-norm clip, AdamW" % args.audio_seconds) if not full else (
                "ONE-PEACE-4B STAGE-2 audio-language pretraining step, objective of pretrain_al_3B.yaml (audio %.0fs): frozen text teacher and "
                "joint al teacher (no grad), audio pass, masked audio / al students (mask ratios .55/.4/.45) through the 2-layer decoder, "
                "ATC + 3 DCL terms; only the audio adapter, audio FFNs and audio_layer_norm of the encoder train (no weight-gradient GEMM "
                "for frozen parameters), backward, grad-norm clip, AdamW" % args.audio_seconds) if al else (
                "ONE-PEACE-4B image-text pretraining step, full objective of pretrain_vl_3B.yaml: text + image teachers, joint vl "
                "teacher (no grad), masked text / image / vl students (mask ratios .15/.75/.4/.6875) through the 2-layer decoder, "
                "ITC + 4 DCL terms, backward, grad-norm clip, AdamW")
        metric = ("pretrain samples/s (tri-modal global batch) ONE-PEACE-4B" if not full else
                  "EXTRA: stage-2 audio-language pretraining objective (ATC + 3 DCL terms) samples/s ONE-PEACE-4B" if al else
                  "EXTRA: full image-text pretraining objective (ITC + 4 DCL terms) samples/s ONE-PEACE-4B")

    if args.fp8:
        if args.config != 4:
            raise SystemExit("--fp8 is the variant of --config 4")
        from one_peace_amd import ops
        ops.set_fp8_ffn(True, dgrad=not args.fp8_forward_only)  # opt-in: FFN GEMMs (up- / down-projection and their input gradients) on e4m3 operands

    if al:
        model = build_pretrain_al_model(args.layers, device, args.recompute)
    elif full:
        model = build_pretrain_vl_model(args.layers, device, args.recompute)
    else:
        model = build_model(args.layers, device, args.recompute, head=head, image_grid=res // 16,
                            extra=dict(text_bucket_size=256) if micro else None)
    if micro:
        model = model.float()  # CPU: the torch path of the mirrors in fp32
    model = model.train() if train else model.eval()
    if args.skip_dropped:
        model.encoder_wrapper.fusion_model.skip_dropped_branches = True
    nparams = sum(p.numel() for p in model.parameters())
    bkw = dict(text_len=15, dtype=torch.float32) if micro else {}
    batch, audio_S = synthetic_batch(args.batch, device, 3407 + rank, res=res, audio_seconds=audio_s, text=args.config != 1, **bkw)
    if full:
        batch = (add_pretrain_al_masks if al else add_pretrain_masks)(batch, 3407 + rank)
    sample = {"net_input": batch, "nsentences": args.batch}

    reducer = opt = flat = None
    if train:
        no_decay_names = model.no_weight_decay()
        flat = FlatParameters(model, no_decay=lambda n, p: p.dim() <= 1 or n in no_decay_names)
        if dist.is_initialized():  # replicas start from rank 0's weights (fairseq: distributed_utils.broadcast of the initial state)
            dist.broadcast(flat.params, src=0)
        reducer = BucketedGradReducer(flat, bucket_bytes=(64 << 10) if micro else (256 << 20))
        opt = (TorchAdamW if micro else FusedAdamW)(flat, lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05)  # pretrain_vl_3B.yaml:24-36
        if al:
            from one_peace_amd.criterions.pretrain import AudioTextPretrainLossCriterion
            crit = AudioTextPretrainLossCriterion(None, 1.0, 0.5, 0.5, 2.5, 0.1)  # pretrain_al_3B.yaml criterion block
        elif full:
            from one_peace_amd.criterions.pretrain import ImageTextPretrainLossCriterion
            crit = ImageTextPretrainLossCriterion(None, 0.5, 1.0, 0.5, 0.5, 2.5, 0.0)  # pretrain_vl_3B.yaml criterion block
        elif args.config == 3:
            crit = TriModalContrastiveCriterion(None, 0.0, lock_step=not args.no_lock_step)
        else:
            crit = ImageTextRetrievalCriterion(None, 0.0, lock_step=not args.no_lock_step)

    feeder = None
    if args.host_inputs:  # fp32 pixels / waveforms on the host, as the reference's collate_fn delivers them
        from one_peace_amd.staging import SamplePrefetcher
        host = {"net_input": {k: (v.float().cpu() if v.is_floating_point() else v.cpu()) for k, v in batch.items()},
                "nsentences": args.batch}

        def forever():
            while True:
                yield host
        feeder = SamplePrefetcher(forever(), device)

    def train_step():
        nonlocal sample
        if feeder is not None:
            sample = next(feeder)
        opt.zero_grad()
        reducer.reset()
        loss, _, log = crit(model, sample)
        loss.backward()
        reducer.finish()
        opt.step(grad_scale=1.0 / world, clip_norm=3.0)  # pretrain_vl_3B.yaml:45 clip_norm; trainer.py:917-935
        return loss

    @torch.no_grad()
    def infer_step(inp=None):
        return model(src_images=(inp if inp is not None else batch["src_images"]), encoder_type="image")

    graph = None

    def graphed_train_step():
        loss = graph.replay()
        reducer.finish()
        opt.step(grad_scale=1.0 / world, clip_norm=3.0)
        return loss

    step = train_step if train else infer_step

    def sync():
        if dist.is_initialized():
            dist.barrier()
        if not micro:
            torch.cuda.synchronize()

    mem_report = {}  # first contact with a multi-GPU node: does the batch survive RCCL's buffers? (config.distributed.memory)

    def fit_batch_to_free_memory():
        """First-contact safety: after the model, the optimiser state and RCCL's own buffers exist, compare what is actually
        free on the device (minimum over ranks) with the activation estimate and halve the batch until it fits -- every rank
        takes the same decision, so no rank can run out of memory in the middle of a collective."""
        nonlocal batch, sample, audio_S
        if not train or micro:
            return
        free_b, _ = torch.cuda.mem_get_info(device)
        free_t = torch.tensor([free_b / 1e9], dtype=torch.float64, device=device)
        if dist.is_initialized():
            dist.all_reduce(free_t, op=dist.ReduceOp.MIN)
        mem_report["free_gb_after_model_optimizer_collectives_min_over_ranks"] = float(free_t.item())
        free_gb = float(free_t.item()) + torch.cuda.memory_reserved(device) / 1e9 - torch.cuda.memory_allocated(device) / 1e9
        changed = False
        passes = 5.0 / 3.0 if full else 1.0  # the full objective keeps five passes (two teachers, three students) instead of ~three

        def kept_gb():  # 1.64 GB per 571-token tuple kept in full, 32 / 46 of it at the cheap level, 0.25 under checkpoint_activations
            per_tuple = 0.25 if args.recompute else 1.64 * 32.0 / 46.0 if args.recompute_cheap else 1.64
            return per_tuple / 571 * 1.0737 * args.layers / LAYERS * sum(modal.values()) * args.batch * passes
        # cushion: 4 GB + the 7.8 GB of transposed dgrad weights the first backward allocates (ops._transposed); world 1 at b = 128
        # measures 287.5 GB peak reserved of 309.2 (config.memory), i.e. ~20 GB really are left for whatever RCCL adds at world > 1
        while args.batch > 1 and kept_gb() > free_gb - 12.0:
            if not args.recompute and not args.recompute_cheap and not args.fp8 and not args.graphs:
                args.recompute_cheap = True  # first resort: the same batch with 30 % fewer kept bytes (+~5 % step time; halving costs ~15 %)
                print("bench: %.0f GB free after model + optimiser + collectives set-up: LayerNorm outputs are re-created in backward "
                      "(--recompute-cheap) instead of kept" % free_gb, file=sys.stderr, flush=True)
                continue
            args.batch //= 2
            changed = True
        if args.recompute_cheap:
            from one_peace_amd import ops
            ops.set_recompute_cheap(True)
        if changed:
            print("bench: %.0f GB free after model + optimiser + collectives set-up, per-GPU batch reduced to %d" % (
                free_gb, args.batch), file=sys.stderr, flush=True)
            batch, audio_S = synthetic_batch(args.batch, device, 3407 + rank, res=res, audio_seconds=audio_s, text=True)
            if full:
                batch = (add_pretrain_al_masks if al else add_pretrain_masks)(batch, 3407 + rank)
            sample = {"net_input": batch, "nsentences": args.batch}

    def warm(n):
        """Warm-up steps.  Single process only: a batch that still runs out of memory is halved once instead of failing the run
        (with several ranks the check above is the safeguard: an exception on one rank would strand the others in a collective)."""
        nonlocal batch, sample, audio_S
        for attempt in range(2):
            try:
                out = None
                for _ in range(n):
                    out = step()
                return out
            except torch.OutOfMemoryError:
                if attempt or not train or world > 1:
                    raise
                if args.recompute_cheap and not args.recompute:  # the cheap level did not fit after all: the reference's level, same batch
                    args.recompute = True
                    for mod in model.modules():
                        if hasattr(getattr(mod, "cfg", None), "checkpoint_activations"):
                            mod.cfg.checkpoint_activations = True
                    model.zero_grad(set_to_none=False)
                    torch.cuda.empty_cache()
                    print("bench: warm-up ran out of memory at the cheap recompute level: layers recompute entirely", file=sys.stderr, flush=True)
                    continue
                args.batch //= 2
                model.zero_grad(set_to_none=False)
                torch.cuda.empty_cache()
                batch, audio_S = synthetic_batch(args.batch, device, 3407 + rank, res=res, audio_seconds=audio_s, text=True)
                if full:
                    batch = (add_pretrain_al_masks if al else add_pretrain_masks)(batch, 3407 + rank)
                sample = {"net_input": batch, "nsentences": args.batch}
                print("bench: warm-up ran out of memory, per-GPU batch halved to %d" % args.batch, file=sys.stderr, flush=True)

    fit_batch_to_free_memory()
    sweep = None
    if args.config == 1:  # the other batch sizes configs[1] names, outside the timed region
        sweep = {}
        for b in (1, 8):
            x = batch["src_images"][:b].contiguous()
            for _ in range(2):
                infer_step(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                infer_step(x)
            torch.cuda.synchronize()
            sweep["batch_%d_images_per_s" % b] = round(b * 5 / (time.perf_counter() - t0), 1)

    # --graphs: no eager step before the capture -- autograd creates the parameters' AccumulateGrad nodes on the stream of the
    # first backward, and nodes that live on the default stream cannot take part in a capture on another one
    loss = warm(max(args.warmup, 0)) if not args.graphs else None
    sync()
    if args.graphs:
        if not train or world > 1 or feeder is not None:
            raise SystemExit("--graphs: single-process training configs with resident inputs")
        from one_peace_amd.graphs import TrainStepGraph
        args.no_profile = True

        def fwd_bwd():
            opt.zero_grad()
            reducer.reset()
            loss, _, _ = crit(model, sample)
            loss.backward()
            return loss
        graph = TrainStepGraph(fwd_bwd, warmup=1)
        step = graphed_train_step
        for _ in range(max(args.warmup, 1)):
            loss = step()
        sync()
    hip.GEMM_ALGO_BYTES[0] = hip.GEMM_ALGO_BYTES[1] = 0
    profiled_steps = 0
    curve = []
    if not args.no_profile:
        hip.lib().op_prof_reserve(8192)  # (the event pairs of one profiled step, created outside the timed region)
    t0 = time.perf_counter()
    for i in range(args.steps):
        prof_on = not args.no_profile and i % PROFILE_EVERY == 0
        if prof_on:
            hip.lib().op_prof_enable(1)
            profiled_steps += 1
        loss = step()
        if args.loss_curve and train:
            curve.append(loss.detach().float().mean())  # (device scalars: read after the timed region)
        if prof_on:
            hip.lib().op_prof_enable(0)
    sync()
    dt = time.perf_counter() - t0
    prof = hip.profile_kernels.collect(4) if profiled_steps else None
    algo_bytes_timed = hip.GEMM_ALGO_BYTES[0]  # (snapshot: the skip leg below launches GEMMs too -- rounds 3-4 counted its bytes in)
    skip_leg = None
    fusion = getattr(getattr(model, "encoder_wrapper", None), "fusion_model", None)
    if (train and world == 1 and not micro and args.config == 3 and not full and not args.skip_dropped and not args.no_skip_leg
            and graph is None and fusion is not None and not args.recompute):
        # second leg, outside `value`: the same step with every residual branch computed for the samples stochastic depth keeps only.
        # It must never cost the run its headline line: any failure (memory: the packed leg allocates other sizes than the first) is reported
        # in place of the figures.
        try:
            fusion.skip_dropped_branches = True
            torch.cuda.empty_cache()  # do not stack the packed leg's sizes on the first leg's cached blocks
            for _ in range(2):
                step()
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                loss2 = step()
            sync()
            dt2 = time.perf_counter() - t1
            skip_leg = {"ms_per_step": dt2 / args.steps * 1e3, "value": args.batch * args.steps / dt2, "unit": "samples/s",
                        "steps": args.steps, "warmup": 2, "final_loss": float(loss2.float().mean().item()),
                        "what": "TransformerEncoder.skip_dropped_branches: every residual branch runs on the packed rows of the samples its "
                                "drop-path mask keeps (op_rows_gather / op_rows_merge); the reference computes all samples and multiplies the "
                                "dropped ones by zero (transformer_layer.py:78-88) -- `value` above does the same.  drop_path_rate 0.4 over "
                                "linspace(0, 0.4, 40): a fifth of the branch work on average"}
        except Exception as e:  # noqa: BLE001
            skip_leg = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            print("bench: the skip_dropped_branches leg failed (%s); the headline line is unaffected" % skip_leg["error"], file=sys.stderr, flush=True)
        finally:
            fusion.skip_dropped_branches = False
            torch.cuda.empty_cache()
    order_same = None
    if args.check_replicas and world > 1:
        chk = torch.stack([flat.params.double().sum(), flat.params.double().abs().sum(), opt.exp_avg.double().sum()])
        allc = [torch.empty_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        same = all(torch.equal(allc[0], c) for c in allc[1:])
        # collectives pair up by issue order: every rank must have issued its bucket all-reduces in the same order
        dig = torch.tensor([reducer.launch_order_digest(), len(reducer.launch_order)], dtype=torch.int64, device=device)
        alld = [torch.empty_like(dig) for _ in range(world)]
        dist.all_gather(alld, dig)
        order_same = all(torch.equal(alld[0], d) for d in alld[1:])
        if rank == 0:
            print("replica check: %s (param checksums %s); bucket launch order on every rank: %s (%d launches)" % (
                "IDENTICAL" if same else "DIVERGED", [c.tolist() for c in allc], "IDENTICAL" if order_same else "DIFFERENT",
                int(alld[0][1])), file=sys.stderr, flush=True)
        assert order_same, "ranks issued their gradient-bucket all-reduces in different orders: %s" % [d.tolist() for d in alld]
        assert same, "data-parallel replicas diverged: a gradient bucket was reduced before its last contribution"
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    loss_v = float(loss.float().mean().item())
    allr = None
    if dist.is_initialized() and not micro:  # memory figures of EVERY rank (a collective: all ranks, not inside the rank-0 report)
        mine = [torch.cuda.max_memory_allocated(device) / 1e9, torch.cuda.max_memory_reserved(device) / 1e9,
                torch.cuda.mem_get_info(device)[0] / 1e9]
        allr = [None] * dist.get_world_size()
        dist.all_gather_object(allr, mine)  # (object collective: gloo has no all_gather of device tensors)

    if rank == 0:
        ms = dt / args.steps * 1e3
        global_batch = args.batch * world
        fwd = sum(fwd_flops_per_sample(S, args.layers) for S in modal.values())
        if "audio" in modal:
            fwd += audio_adapter_fwd_flops(args.audio_seconds)
        fl = (3.0 if train else 1.0) * fwd
        out = {
            "metric": metric, "value": global_batch * args.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" + (" (FFN GEMMs: fp8 e4m3 operands, fp32 accumulate)" if args.fp8 else ""),
            "data": "synthetic (random-init weights, seeded random tokens / N(0,1) pixels and waveforms)"
                    + ("; inputs staged from pinned host memory every step" if args.host_inputs else ""),
            "config": {"workload": name, "baseline_config_index": args.config,
                       "embed_dim": H, "ffn": FFN, "layers": args.layers, "heads": HEADS, "params": nparams,
                       "per_gpu_batch": args.batch, "global_batch": global_batch, "tokens_per_sample": sum(modal.values()),
                       "sequence_lengths": modal, "parallelism": "dp%d" % world,
                       "collectives": ("%s: broadcast, [k,b,H] all-gather, bucketed gradient all-reduce" % dist.get_backend()
                                       if dist.is_initialized() else "none (single process)"),
                       "activation_recompute": ("n/a (no-grad forward)" if not train else
                                                "per layer (the reference's checkpoint_activations: true)" if args.recompute else
                                                "cheap level: layer activations kept in HBM except the four LayerNorm-type outputs only weight "
                                                "gradients read, which backward re-creates (ops.set_recompute_cheap)" if args.recompute_cheap
                                                else "off: layer activations are kept in HBM (288 GB/GPU)"),
                       "algorithmic_tflop_per_sample": None if full else fl / 1e12,
                       "step_algorithmic_tflops_per_gpu": None if full else fl * args.batch / (ms / 1e3) / 1e12,
                       "objective": "forward" if not train else args.objective, "final_loss": loss_v if train else None,
                       **({"param_checksum": param_checksum(flat.params),
                           "nt_gemm_launch_rule": ("one tile per workgroup (tune sched 7: distributed.share_cus_with_collectives)"
                                                   if hip.TUNE.sched == 7 else "persistent workgroups for K <= 2048 and grouped launches"
                                                   if hip.TUNE.sched == 0 else "tune sched %d" % hip.TUNE.sched)} if train and not micro else {}),
                       **({"loss_curve": [round(float(x), 5) for x in curve]} if curve else {}),
                       "launch_path": ("hipGraph replay of zero-grad + forwards + loss + backward, eager optimiser step" if args.graphs
                                       else "eager (one ctypes call per kernel)")},
        }
        if sweep:
            out["config"]["sweep"] = sweep
        if not micro and world == 1:
            out["config"]["memory"] = {"peak_allocated_gb": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
                                       "peak_reserved_gb": round(torch.cuda.max_memory_reserved(device) / 1e9, 2),
                                       "device_total_gb": round(torch.cuda.mem_get_info(device)[1] / 1e9, 2)}
        if train:
            out["config"]["stochastic_depth"] = ("branches of dropped samples are not computed (--skip-dropped)" if args.skip_dropped
                                                 else "reference arithmetic: every sample computed, dropped ones multiplied by zero")
        if skip_leg is not None:
            out["skip_dropped_branches"] = skip_leg
        if micro:
            out["metric"] = "NOT A MEASUREMENT: CPU control-flow run of the data-parallel step (micro model, gloo)"
            out["dtype"], out["roofline"] = "f32 (torch path on CPU)", None
        if dist.is_initialized():
            k_mod = len(modal)
            out["config"]["distributed"] = {
                "backend": dist.get_backend(), "world_size_seen_by_backend": dist.get_world_size(),
                "grad_buckets": len(reducer.buckets) if reducer is not None else 0,
                "grad_bucket_bytes": [(e - s0) * flat.grads.element_size() for s0, e, _ in reducer.buckets][:4] if reducer is not None else [],
                "grad_allreduce_bytes_per_step": flat.numel * flat.grads.element_size() if flat is not None else 0,
                "embedding_allgather_bytes_per_step": k_mod * args.batch * H * (4 if micro else 2) * dist.get_world_size(),
                "bucket_launch_order_identical_on_all_ranks": order_same}
            if allr is not None:  # per-rank peak memory and what was left of the device at the peak (the 6 GB RCCL reserve has never met hardware)
                total_b = torch.cuda.mem_get_info(device)[1]
                mem_report.update({"device_total_gb": total_b / 1e9,
                                   "peak_allocated_gb_per_rank": [round(float(t[0]), 2) for t in allr],
                                   "peak_reserved_gb_per_rank": [round(float(t[1]), 2) for t in allr],
                                   "free_gb_at_exit_per_rank": [round(float(t[2]), 2) for t in allr],
                                   "margin_gb_min_over_ranks": round(min(total_b / 1e9 - float(t[1]) for t in allr), 2)})
                out["config"]["distributed"]["memory"] = mem_report
                print("bench: memory margin (device total - peak reserved), minimum over %d ranks: %.2f GB; per-GPU batch %d" % (
                    len(allr), mem_report["margin_gb_min_over_ranks"], args.batch), file=sys.stderr, flush=True)
        if reducer is not None and (world > 1 or reducer.active):
            rep = reducer.overlap_report()
            out["config"]["grad_allreduce_overlap"] = rep
            # every parameter of this model takes part in every step: a bucket that only goes out in finish() means a
            # completion signal was lost (the three passes share the attention weights) -- fail loudly, never silently serialise
            if rep["launched_in_finish"] > rep["buckets"]:  # (the first step learns which parameters are unused: its buckets may wait)
                msg = "gradient buckets were not all all-reduced during backward (no overlap for them): %s" % rep
                assert not args.check_replicas, msg
                print("bench: WARNING " + msg, file=sys.stderr, flush=True)
        if prof is not None and prof[0]["count"] > 0:
            g = prof[0]
            ach = g["work"] / (g["ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma",
                               "kernel": "GEMM family (gemm256v / gemm256p four-wave NT kernels, the grouped weight-gradient launch gemm256w_tn_grouped + "
                                         "gemm256_tn / gemm256w_tn for row counts it does not take, gemm_nt tail-rows kernel, incl. split-K folds)",
                               "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                               "traffic": None, "launches": g["count"], "avg_launch_ms": g["ms"] / g["count"],
                               "profiled_steps": profiled_steps,
                               "algorithmic_bytes_per_launch": algo_bytes_timed * profiled_steps / args.steps / max(1, g["count"]),
                               "gemm_share_of_step": g["ms"] / (ms * profiled_steps),
                               "attention_fwd_tflops": (prof[1]["work"] / (prof[1]["ms"] * 1e-3) / 1e12) if prof[1]["count"] else None,
                               "attention_bwd_tflops": (prof[2]["work"] / (prof[2]["ms"] * 1e-3) / 1e12) if prof[2]["count"] else None}
            if prof[3]["count"] > 0:  # fp8 GEMMs are their own family with their own peak
                f8 = prof[3]
                a8 = f8["work"] / (f8["ms"] * 1e-3) / 1e12
                out["roofline"]["fp8_gemm"] = {"achieved": a8, "peak": PEAK_FP8_TFLOPS, "frac": a8 / PEAK_FP8_TFLOPS,
                                               "launches": f8["count"], "share_of_step": f8["ms"] / (ms * profiled_steps)}
            # HBM bytes per GEMM launch need separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), so they
            # cannot be collected inside this process: tools/pmc_bench_traffic.sh runs this command under those passes and writes
            # the file; it is quoted only for the exact configuration it was measured on.
            for tname in ("r6_gemm_hbm_traffic.json", "r5_gemm_hbm_traffic.json", "r4_gemm_hbm_traffic.json", "r3_gemm_hbm_traffic.json", "r2_gemm_hbm_traffic.json", "r1_gemm_hbm_traffic.json"):
                tpath = os.path.join(ROOT, "profiles", tname)
                if not os.path.exists(tpath):
                    continue
                tr = json.load(open(tpath))
                if (tr.get("per_gpu_batch") == args.batch and tr.get("n_gpus") == world and args.layers == LAYERS
                        and tr.get("config", 3) == args.config and not full and not args.fp8):
                    out["roofline"]["traffic"] = tr["bytes_per_launch"]
                    out["roofline"]["traffic_source"] = ("profiles/%s: committed rocprofv3 --pmc passes of this command "
                                                         "(not collected in this run)" % tname)
                    break
        if world == 1 and out.get("roofline") and not args.no_power_probe:
            # `peak` above is the data sheet's dense-bf16 figure at 2.4 GHz.  With random operands the package reaches its 1400 W
            # limit long before that clock: a register-only MFMA loop (op_probe_mfma_rate, ~1 s, after the timed region) measures
            # what the limit leaves on THIS box, and the GEMM family is quoted against that as well (DESIGN.md "Power").
            try:
                pp = hip.mfma_rate_probe(seconds=1.0, waves_per_cu=8, data="normal")
                out["roofline"]["power_limited_peak"] = {
                    "tflops": pp["tflops"], "shader_mhz": pp["mhz"], "frac": out["roofline"]["achieved"] / pp["tflops"],
                    "how": "register-only v_mfma_f32_16x16x32_bf16 loop, 8 waves per CU, random normal operands, %.0f ms, measured in this run"
                           % pp["ms"]}
            except Exception as e:  # noqa: BLE001  (an extra: never at the cost of the line)
                out["roofline"]["power_limited_peak"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(modal, backward=train)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
