"""Mirrors of one_peace/criterions/{image_text,audio_text}_pretrain_loss.py: the full pretraining objective
ITC/ATC + four masked-token (DCL) terms -- six forward passes per step: two teacher passes that also give the contrastive
logits, one joint teacher pass without gradient, and three student passes whose adapters keep a different token subset
per sample (``*_preserve_ids``) and whose features go through the small decoder.

On an MI355X every pass runs on the fused HIP layers (the masked ones with per-sample bias images), and the DCL
similarity matrix is never materialised (``ops.dcl_loss``: blockwise GEMM -> InfoNCE rows -> GEMM); elsewhere the torch ops
written below compute the same values."""
import torch
import torch.nn.functional as F

from .. import ops
from ..registry import FairseqCriterion, register_criterion
from .contrastive import adjust_label_smoothed_nll_loss, contrastive_pair_loss, gather_without_grad


def compute_dcl_loss(student_features, teacher_features, mask_indices, dcl_logit_scale, label_smoothing=0.0,
                     padding_masks=None):
    """image_text_pretrain_loss.py:187-208."""
    H = student_features.size(-1)
    student = student_features[:, 1:, :].reshape(-1, H)
    teacher = teacher_features.detach()[:, 1:, :].reshape(-1, H)
    mask = mask_indices[:, 1:].flatten()
    if padding_masks is not None:
        keep = torch.nonzero((~padding_masks).flatten(), as_tuple=False).flatten()
        student, teacher, mask = student[keep], teacher[keep], mask[keep]
    indices = torch.nonzero(mask, as_tuple=False).flatten()
    if ops.hip_eligible(student) and H % 8 == 0 and indices.numel() > 0:
        rest = torch.nonzero(~mask, as_tuple=False).flatten()
        s_n = ops.l2_normalize(student[indices].contiguous())
        t_n = ops.l2_normalize(teacher[torch.cat([indices, rest])].contiguous())  # masked positions first: target(i) = i
        return ops.dcl_loss(s_n, t_n, dcl_logit_scale, label_smoothing)
    orig = student.dtype
    s_n = F.normalize(student[indices].float(), dim=1).to(orig)
    t_n = F.normalize(teacher.float(), dim=1).to(orig)
    sim = dcl_logit_scale * s_n @ t_n.t()
    lp = F.log_softmax(sim, dim=-1, dtype=torch.float32).type_as(sim)
    return adjust_label_smoothed_nll_loss(lp, indices, label_smoothing)


def _log_base(loss, sample, scale):
    return {"loss": loss.data, "nsentences": sample["nsentences"], "sample_size": 1, "logit_scale_exp": scale.data}


@register_criterion("image_text_pretrain_loss")
class ImageTextPretrainLossCriterion(FairseqCriterion):
    """image_text_pretrain_loss.py:54-160 (constructor argument order of the reference kept)."""

    def __init__(self, task, dcl_text_alpha=0.5, dcl_image_alpha=1.0, dcl_vl_text_alpha=0.5, dcl_vl_image_alpha=0.5,
                 dcl_logit_scale=2.5, label_smoothing=0.0):
        super().__init__(task)
        self.dcl_text_alpha, self.dcl_image_alpha = dcl_text_alpha, dcl_image_alpha
        self.dcl_vl_text_alpha, self.dcl_vl_image_alpha = dcl_vl_text_alpha, dcl_vl_image_alpha
        self.dcl_logit_scale, self.label_smoothing = dcl_logit_scale, label_smoothing
        self.lock_step = True  # False: one model call per pass, exactly the reference's call pattern

    def compute_dcl_loss(self, student_features, teacher_features, mask_indices, padding_masks=None):
        return compute_dcl_loss(student_features, teacher_features, mask_indices, self.dcl_logit_scale, self.label_smoothing,
                                padding_masks)

    def compute_itc_loss(self, image_logits, text_logits, image_logits_all, text_logits_all, logit_scale_exp):
        return contrastive_pair_loss(image_logits, text_logits, image_logits_all, text_logits_all, logit_scale_exp)

    def forward(self, model, sample, reduce=True):
        ni = sample["net_input"]
        tok, img = ni["src_tokens"], ni["src_images"]
        multi = model.forward_multi(src_tokens=tok, src_images=img) if self.lock_step and hasattr(model, "forward_multi") else None
        if multi is not None:  # the two unmasked single-modality passes in lock-step (MI355X path; same rows, same arithmetic)
            (text_logits, teacher_text), (image_logits, teacher_image) = multi["text"], multi["image"]
        else:
            text_logits, teacher_text = model(src_tokens=tok, encoder_type="text")
            image_logits, teacher_image = model(src_images=img, encoder_type="image")
        text_all, image_all = gather_without_grad(text_logits, image_logits)
        with torch.no_grad():
            teacher_vl_text, teacher_vl_image = model(src_tokens=tok, src_images=img, encoder_type="vl")
        student_text, _, _ = model(src_tokens=tok, text_preserve_ids=ni["text_preserve_ids"], encoder_type="text")
        _, student_image, _ = model(src_images=img, image_preserve_ids=ni["image_preserve_ids"], encoder_type="image")
        student_vl_text, student_vl_image, _ = model(src_tokens=tok, text_preserve_ids=ni["vl_text_preserve_ids"], src_images=img,
                                                     image_preserve_ids=ni["vl_image_preserve_ids"], encoder_type="vl")
        scale = model(return_logit_scale=True)
        pads = tok.eq(1)
        l_text = self.compute_dcl_loss(student_text, teacher_text, ni["text_mask_indices"], pads)
        l_image = self.compute_dcl_loss(student_image, teacher_image, ni["image_mask_indices"])
        l_vl_text = self.compute_dcl_loss(student_vl_text, teacher_vl_text, ni["vl_text_mask_indices"], pads)
        l_vl_image = self.compute_dcl_loss(student_vl_image, teacher_vl_image, ni["vl_image_mask_indices"])
        itc, i2t, t2i = self.compute_itc_loss(image_logits, text_logits, image_all, text_all, scale)
        loss = (itc + self.dcl_text_alpha * l_text + self.dcl_image_alpha * l_image
                + self.dcl_vl_text_alpha * l_vl_text + self.dcl_vl_image_alpha * l_vl_image)
        log = _log_base(loss, sample, scale)
        log.update(itc_loss=itc.data, dcl_text_loss=l_text.data, dcl_image_loss=l_image.data, dcl_vl_text_loss=l_vl_text.data,
                   dcl_vl_image_loss=l_vl_image.data, i2t_ncorrect=i2t, t2i_ncorrect=t2i)
        return loss, 1, log

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return True


@register_criterion("audio_text_pretrain_loss")
class AudioTextPretrainLossCriterion(FairseqCriterion):
    """audio_text_pretrain_loss.py:53-157: the audio-language stage -- the text tower is a frozen teacher (its pass runs
    without gradient), both audio DCL terms use the joint 'al' teacher, and there is no text-only DCL term."""

    def __init__(self, task, dcl_audio_alpha=1.0, dcl_al_text_alpha=0.5, dcl_al_audio_alpha=0.5, dcl_logit_scale=2.5,
                 label_smoothing=0.0):
        super().__init__(task)
        self.dcl_audio_alpha, self.dcl_al_text_alpha, self.dcl_al_audio_alpha = dcl_audio_alpha, dcl_al_text_alpha, dcl_al_audio_alpha
        self.dcl_logit_scale, self.label_smoothing = dcl_logit_scale, label_smoothing

    def compute_dcl_loss(self, student_features, teacher_features, mask_indices, padding_masks=None):
        return compute_dcl_loss(student_features, teacher_features, mask_indices, self.dcl_logit_scale, self.label_smoothing,
                                padding_masks)

    def compute_atc_loss(self, audio_logits, text_logits, audio_logits_all, text_logits_all, logit_scale_exp):
        return contrastive_pair_loss(audio_logits, text_logits, audio_logits_all, text_logits_all, logit_scale_exp)

    def forward(self, model, sample, reduce=True):
        ni = sample["net_input"]
        tok, wav, apad = ni["src_tokens"], ni["src_audios"], ni["audio_padding_masks"]
        with torch.no_grad():
            text_logits, _ = model(src_tokens=tok, encoder_type="text")
        audio_logits, _ = model(src_audios=wav, audio_padding_masks=apad, encoder_type="audio")
        text_all, audio_all = gather_without_grad(text_logits, audio_logits)
        with torch.no_grad():
            teacher_al_text, teacher_al_audio = model(src_tokens=tok, src_audios=wav, audio_padding_masks=apad, encoder_type="al")
        _, _, student_audio = model(src_audios=wav, audio_preserve_ids=ni["audio_preserve_ids"], audio_padding_masks=apad,
                                    encoder_type="audio")
        student_al_text, _, student_al_audio = model(src_tokens=tok, text_preserve_ids=ni["al_text_preserve_ids"], src_audios=wav,
                                                     audio_padding_masks=apad, audio_preserve_ids=ni["al_audio_preserve_ids"],
                                                     encoder_type="al")
        scale = model(return_logit_scale=True)
        tpads, apads = tok.eq(1), apad[:, 1:]
        l_audio = self.compute_dcl_loss(student_audio, teacher_al_audio, ni["audio_mask_indices"], apads)
        l_al_text = self.compute_dcl_loss(student_al_text, teacher_al_text, ni["al_text_mask_indices"], tpads)
        l_al_audio = self.compute_dcl_loss(student_al_audio, teacher_al_audio, ni["al_audio_mask_indices"], apads)
        atc, a2t, t2a = self.compute_atc_loss(audio_logits, text_logits, audio_all, text_all, scale)
        loss = atc + self.dcl_audio_alpha * l_audio + self.dcl_al_text_alpha * l_al_text + self.dcl_al_audio_alpha * l_al_audio
        log = _log_base(loss, sample, scale)
        log.update(atc_loss=atc.data, dcl_audio_loss=l_audio.data, dcl_al_text_loss=l_al_text.data,
                   dcl_al_audio_loss=l_al_audio.data, a2t_ncorrect=a2t, t2a_ncorrect=t2a)
        return loss, 1, log

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return True
