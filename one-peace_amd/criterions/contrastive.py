"""Contrastive criteria of the hot path, mirroring
one_peace/criterions/{image_text,audio_text}_retrieval_loss.py and the ITC/ATC part of *_pretrain_loss.py.

``forward(model, sample, reduce=True) -> (loss, sample_size=1, logging_output)`` as in fairseq.  Differences that are
MI355X design, not semantics: the two modality embeddings are exchanged with ONE ``all_gather_into_tensor`` of
``[2, b, H]`` instead of two list-API all-gathers (the gathered copies still carry no gradient, and rank r's rows sit
at ``[r*b, (r+1)*b)``), and sim/log-softmax/NLL/argmax run in the HIP InfoNCE kernels on bf16 device tensors."""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .. import ops
from ..registry import FairseqCriterion, register_criterion

try:
    from fairseq import metrics as _metrics  # type: ignore
except Exception:
    _metrics = None


def adjust_label_smoothed_nll_loss(lprobs, target, epsilon=0.0):
    nll = -lprobs.gather(-1, target.unsqueeze(-1) if target.dim() == lprobs.dim() - 1 else target).squeeze(-1)
    if epsilon != 0:
        eps_i = epsilon / (lprobs.size(-1) - 1)
        nll = (1.0 - epsilon - eps_i) * nll - eps_i * lprobs.sum(dim=-1)
    return nll.mean()


@torch.no_grad()
def gather_without_grad(*tensors):
    """All-gather same-shape tensors in one collective; returns rank-major concatenations (no gradient)."""
    if not (dist.is_available() and dist.is_initialized()):
        return tuple(t.detach() for t in tensors)
    world = dist.get_world_size()
    packed = torch.stack([t.detach() for t in tensors], dim=0).contiguous()  # [k, b, H]
    out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out.view(-1), packed.view(-1))
    # out[r, i] = tensor i of rank r  ->  [k][world*b, H]
    return tuple(out[:, i].reshape(-1, *packed.shape[2:]) for i in range(len(tensors)))


def contrastive_pair_loss(a_local, b_local, a_all, b_all, scale, label_smoothing=0.0):
    """(loss, a->b hits, b->a hits) for the pair (a = image/audio, b = text); targets rank*bsz + i."""
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    if ops.hip_eligible(a_local) and a_local.shape[1] % 8 == 0:
        return ops.info_nce(a_local, b_local, a_all, b_all, scale, rank, label_smoothing)
    bsz = a_local.size(0)
    tgt = torch.arange(bsz, device=a_local.device) + bsz * rank
    sim_ab = scale * a_local @ b_all.t()
    sim_ba = scale * b_local @ a_all.t()
    la = F.log_softmax(sim_ab, dim=-1, dtype=torch.float32).type_as(sim_ab)
    lb = F.log_softmax(sim_ba, dim=-1, dtype=torch.float32).type_as(sim_ba)
    loss = (adjust_label_smoothed_nll_loss(la, tgt, label_smoothing) + adjust_label_smoothed_nll_loss(lb, tgt, label_smoothing)) / 2
    with torch.no_grad():
        a_ok = (sim_ab.argmax(dim=1) == tgt).float().sum()
        b_ok = (sim_ba.argmax(dim=1) == tgt).float().sum()
    return loss, a_ok, b_ok


def _first(x):
    """Pretrain models return (logits, features); retrieval models return logits."""
    return x[0] if isinstance(x, tuple) else x


class _PairCriterion(FairseqCriterion):
    other = "image"          # the non-text modality
    keys = ("i2t", "t2i")

    def __init__(self, task, label_smoothing=0.0, lock_step=True):
        super().__init__(task)
        self.label_smoothing = label_smoothing
        self.lock_step = lock_step  # False: always one model call per modality, exactly the reference's call pattern

    def _other_logits(self, model, net_input):
        raise NotImplementedError

    def _other_inputs(self, net_input):
        raise NotImplementedError

    def forward(self, model, sample, reduce=True):
        ni = sample["net_input"]
        multi = None
        if self.lock_step and hasattr(model, "forward_multi"):  # both streams layer by layer in lock-step (MI355X path; round 4)
            multi = model.forward_multi(src_tokens=ni["src_tokens"], **self._other_inputs(ni))
        if multi is not None:
            text, other = _first(multi["text"]), _first(multi[self.other])  # pretrain models: (logits, features)
        else:
            text = _first(model(src_tokens=ni["src_tokens"], encoder_type="text"))
            other = self._other_logits(model, ni)
        text_all, other_all = gather_without_grad(text, other)
        scale = model(return_logit_scale=True)
        loss, o2t, t2o = contrastive_pair_loss(other, text, other_all, text_all, scale, self.label_smoothing)
        logging_output = {"loss": loss.data, "nsentences": sample["nsentences"], "sample_size": 1,
                          self.keys[0] + "_ncorrect": o2t, self.keys[1] + "_ncorrect": t2o, "logit_scale_exp": scale.data}
        return loss, 1, logging_output

    @classmethod
    def reduce_metrics(cls, logging_outputs) -> None:
        if _metrics is None:
            return
        tot = lambda k, d=0: sum(log.get(k, d) for log in logging_outputs)  # noqa: E731
        sample_size, nsent = tot("sample_size", 1), tot("nsentences", 1)
        _metrics.log_scalar("loss", tot("loss") / sample_size, sample_size, round=3)
        _metrics.log_scalar("logit_scale_exp", tot("logit_scale_exp") / sample_size, sample_size, round=3)
        _metrics.log_scalar("nsentences", nsent, 1, round=3)
        _metrics.log_scalar("sample_size", sample_size, 1, round=3)
        for k in cls.keys:
            if logging_outputs and (k + "_ncorrect") in logging_outputs[0]:
                _metrics.log_scalar(k + "_accuracy", 100.0 * tot(k + "_ncorrect") / nsent, nsent, round=1)

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return True


@register_criterion("image_text_retrieval_criterion")
class ImageTextRetrievalCriterion(_PairCriterion):
    other, keys = "image", ("i2t", "t2i")

    def _other_logits(self, model, ni):
        return _first(model(src_images=ni["src_images"], encoder_type="image"))

    def _other_inputs(self, ni):
        return {"src_images": ni["src_images"]}

    def compute_itc_loss(self, image_logits, text_logits, image_logits_all, text_logits_all, logit_scale_exp):
        return contrastive_pair_loss(image_logits, text_logits, image_logits_all, text_logits_all, logit_scale_exp,
                                     self.label_smoothing)


@register_criterion("audio_text_retrieval_criterion")
class AudioTextRetrievalCriterion(_PairCriterion):
    other, keys = "audio", ("a2t", "t2a")

    def _other_logits(self, model, ni):
        return _first(model(src_audios=ni["src_audios"], audio_padding_masks=ni["audio_padding_masks"],
                            encoder_type="audio"))

    def _other_inputs(self, ni):
        return {"src_audios": ni["src_audios"], "audio_padding_masks": ni["audio_padding_masks"]}

    def compute_atc_loss(self, audio_logits, text_logits, audio_logits_all, text_logits_all, logit_scale_exp):
        return contrastive_pair_loss(audio_logits, text_logits, audio_logits_all, text_logits_all, logit_scale_exp,
                                     self.label_smoothing)


@register_criterion("tri_modal_contrastive_criterion")
class TriModalContrastiveCriterion(FairseqCriterion):
    """BASELINE config 4: one step over (image, audio, text) tuples = three single-modality forwards + ITC(image, text)
    + ATC(audio, text).  The reference has no joint 3-way criterion ('val' is unimplemented,
    transformer_encoder.py:136-137): VL and AL are separate stages with exactly these two losses, which this criterion
    sums so that all three FFN sets train in one step.  One fused all-gather of [3, b, H]."""

    def __init__(self, task, label_smoothing=0.0, lock_step=True):
        super().__init__(task)
        self.label_smoothing = label_smoothing
        self.lock_step = lock_step  # False: always one model call per modality, exactly the reference's call pattern

    def forward(self, model, sample, reduce=True):
        ni = sample["net_input"]
        multi = None
        if self.lock_step and hasattr(model, "forward_multi"):  # the three streams layer by layer in lock-step (MI355X path)
            multi = model.forward_multi(src_tokens=ni["src_tokens"], src_images=ni["src_images"], src_audios=ni["src_audios"],
                                        audio_padding_masks=ni["audio_padding_masks"])
        if multi is not None:
            text, image, audio = (_first(multi[m]) for m in ("text", "image", "audio"))
        else:
            text = _first(model(src_tokens=ni["src_tokens"], encoder_type="text"))
            image = _first(model(src_images=ni["src_images"], encoder_type="image"))
            audio = _first(model(src_audios=ni["src_audios"], audio_padding_masks=ni["audio_padding_masks"], encoder_type="audio"))
        text_all, image_all, audio_all = gather_without_grad(text, image, audio)
        scale = model(return_logit_scale=True)
        itc, i2t, t2i = contrastive_pair_loss(image, text, image_all, text_all, scale, self.label_smoothing)
        atc, a2t, t2a = contrastive_pair_loss(audio, text, audio_all, text_all, scale, self.label_smoothing)
        loss = itc + atc
        log = {"loss": loss.data, "itc_loss": itc.data, "atc_loss": atc.data, "nsentences": sample["nsentences"],
               "sample_size": 1, "i2t_ncorrect": i2t, "t2i_ncorrect": t2i, "a2t_ncorrect": a2t, "t2a_ncorrect": t2a,
               "logit_scale_exp": scale.data}
        return loss, 1, log

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return True
