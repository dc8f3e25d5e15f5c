"""Host-side mirror of the reference's model interface for the inference hot path.

``VALLE`` keeps the constructor and the ``inference`` signature of
``valle.models.VALLE`` (/root/reference/valle/models/valle.py:727-760, 961-985) and
``get_model`` / ``add_model_arguments`` those of /root/reference/valle/models/__init__.py, so
that ``valle/bin/infer.py`` only needs its import line changed.  All arithmetic
runs in libvallex.so through the C ABI (engine.py); this file holds only what the reference
does on the host: argument checks, the prefix-mode text trim, RNG bookkeeping, exceptions.
"""
from __future__ import annotations

from collections import OrderedDict, namedtuple
from typing import Optional

import torch

from .config import NUM_AUDIO_TOKENS, NUM_TEXT_TOKENS, ModelConfig, add_model_arguments  # noqa: F401
from .weights import expected_keys, synthetic_state_dict, tied_keys

_IncompatibleKeys = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])


def _ids_out_of_range(x: torch.Tensor, y: torch.Tensor) -> bool:
    """Text ids outside [0, NUM_TEXT_TOKENS) or codec ids outside [0, NUM_AUDIO_TOKENS): both tests on the device, ONE
    device-to-host read (four `int(t.min())`-style syncs before).  An empty prompt is legal with prepend_bos."""
    tx, ty = x.reshape(-1), y.reshape(-1)
    flags = [((tx < 0) | (tx >= NUM_TEXT_TOKENS)).any()]
    if ty.numel():
        flags.append(((ty < 0) | (ty >= NUM_AUDIO_TOKENS)).any())
    return bool(torch.stack(flags).any())


class VALLE:
    """Decoder-only VALL-E (inference only).  Engine-specific keyword arguments (not in the
    reference): ``precision`` ("bf16" | "fp32"), ``max_text``, ``max_audio`` (capacities),
    ``sampling`` ("device": on-GPU counter RNG seeded from torch's global generator;
    "torch_cpu": reproduce the exact Exp(1) stream torch.multinomial would consume on CPU)."""

    MODEL_NAME = "VALL-E"  # what the EOS line prints and get_model dispatches on (models/__init__.py:98-124)

    def __init__(self, d_model: int, nhead: int, num_layers: int, norm_first: bool = True, add_prenet: bool = False,
                 prefix_mode: int = 0, share_embedding: bool = True, nar_scale_factor: float = 1.0, **kwargs):
        self.engine_opts = dict(
            precision=kwargs.pop("precision", "bf16"), max_text=kwargs.pop("max_text", 512),
            max_audio=kwargs.pop("max_audio", 4096), trace_logits=kwargs.pop("trace_logits", False),
            no_graph=kwargs.pop("no_graph", False), simple_rows=kwargs.pop("simple_rows", False),
            max_batch=kwargs.pop("max_batch", 0))
        self.sampling = kwargs.pop("sampling", "device")
        self.print_eos = kwargs.pop("print_eos", True)
        self.cfg = ModelConfig(model_name=self.MODEL_NAME, decoder_dim=d_model, nhead=nhead, num_decoder_layers=num_layers, norm_first=norm_first,
                               add_prenet=add_prenet, prefix_mode=prefix_mode, share_embedding=share_embedding,
                               scale_factor=nar_scale_factor, prepend_bos=kwargs.pop("prepend_bos", False),
                               num_quantizers=kwargs.pop("num_quantizers", 8))
        if kwargs:
            raise TypeError(f"unexpected arguments {sorted(kwargs)}")
        if (not norm_first or add_prenet or self.cfg.is_vallf) and self.engine_opts.get("max_batch", 0) > 1:
            raise NotImplementedError("norm_first=False / add_prenet=True / VALL-F run on the batch-1 path only (inference_batch needs the defaults)")
        if self.cfg.is_vallf and self.engine_opts["precision"] == "fp8nar":
            raise NotImplementedError("precision 'fp8nar' is built for VALL-E only")
        # head_dim 64 is the tuned geometry; 4/8/16/32 (the reference's own test: decoder_dim 64, nhead 16, valle_test.py:93-95)
        # run on the plain kernels, batch-1 only
        hds = [d_model // nhead if nhead > 0 and d_model % nhead == 0 else 0]
        if self.cfg.num_quantizers > 1:
            hds.append(self.cfg.nar_dim // self.cfg.nar_nhead if self.cfg.nar_nhead > 0 and self.cfg.nar_dim % self.cfg.nar_nhead == 0 else 0)
        if any(h not in (4, 8, 16, 32, 64) for h in hds):
            raise NotImplementedError(f"head_dim must be 4, 8, 16, 32 or 64 (got {hds}; DESIGN.md)")
        if any(h != 64 for h in hds) and self.engine_opts.get("max_batch", 0) > 1:
            raise NotImplementedError("inference_batch needs head_dim 64")
        self.ar_audio_prepend_bos = self.cfg.prepend_bos
        self.num_quantizers = self.cfg.num_quantizers
        self.prefix_mode = prefix_mode
        self.num_heads = nhead
        self.device = torch.device("cpu")
        self.training = True
        self._engine = None
        # nn.Module would random-init here; same distributions, fresh seed from the global RNG
        self._sd = synthetic_state_dict(self.cfg, seed=int(torch.randint(0, 2**31 - 1, (1,))), zero_eos=False)

    # ---- nn.Module-like surface used by bin/infer.py:137-148 ---------------------------------------
    def state_dict(self):
        return OrderedDict(self._sd)

    def load_state_dict(self, state_dict, strict: bool = True):
        want = expected_keys(self.cfg)
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        bad = [k for k in want if k in state_dict and tuple(state_dict[k].shape) != tuple(want[k])]
        if bad or (strict and (missing or unexpected)):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: missing {missing}, unexpected {unexpected}, "
                               f"size mismatch {bad}")
        for k in want:
            if k in state_dict:
                keep = k.endswith("num_batches_tracked")  # int64 scalar of BatchNorm1d
                self._sd[k] = state_dict[k].detach().to("cpu", state_dict[k].dtype if keep else torch.float32).contiguous()
        if self.cfg.share_embedding:  # valle.py:261-271: tied tensors are one storage
            for pk, ek in tied_keys(self.cfg).items():
                if pk in state_dict and ek in state_dict and not torch.equal(self._sd[pk], self._sd[ek]):
                    raise RuntimeError(f"tied weights differ: {pk} vs {ek}")
        self._drop_engine()
        return _IncompatibleKeys(missing, unexpected)

    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def to(self, device):
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._drop_engine()
        return self

    def cuda(self, index: int = 0):
        return self.to(torch.device("cuda", index))

    def eval(self):
        self.training = False
        return self

    def engine(self):
        """Creates the HIP engine on first use; there is no CPU path."""
        if self._engine is None:
            if self.device.type != "cuda":
                raise RuntimeError("valle_amd.VALLE runs only on an MI355X: call .to('cuda') first (no CPU fallback)")
            from .engine import Engine

            self._engine = Engine(self.cfg, device=self.device.index or 0, **self.engine_opts)
            self._engine.load_state_dict(self._sd)
        return self._engine

    # ---- the hot path -------------------------------------------------------------------------------
    @torch.no_grad()
    def inference(self, x: torch.Tensor, x_lens: torch.Tensor, y: torch.Tensor, enroll_x_lens: Optional[torch.Tensor],
                  top_k: int = -100, temperature: float = 1.0, exp_noise: Optional[torch.Tensor] = None,
                  max_new_tokens: int = -1) -> torch.Tensor:
        """Same contract as the reference (valle.py:961-985): x (1,S) int64, x_lens (1,), y (1,P,8) int64 →
        (1,T,num_quantizers) int64 on the model's device.  ``exp_noise`` / ``max_new_tokens`` are extras."""
        assert x.ndim == 2, x.shape  # valle.py:986-991
        assert x_lens.ndim == 1, x_lens.shape
        assert y.ndim == 3, y.shape
        assert y.shape[0] == 1, y.shape
        assert torch.all(x_lens > 0)
        S = int(x_lens.max())
        if x.shape[1] != S or x.shape[0] != 1:
            # the reference builds its mask from x_lens.max() but concatenates all of x: any padding makes
            # its attention shapes disagree (valle.py:1009-1038)
            raise RuntimeError(f"x must be one unpadded sequence: x {tuple(x.shape)} vs x_lens.max() {S}")
        eng = self.engine()
        Q, bos = self.num_quantizers, int(self.ar_audio_prepend_bos)
        if _ids_out_of_range(x, y[..., :Q]):
            raise IndexError("index out of range in self")  # what nn.Embedding raises in the reference
        text = x[0]
        prompts = y[0, :, :Q].contiguous()
        P = prompts.shape[0]

        eng.ar_prefill(text, prompts[:, 0].contiguous())
        n_max = max(1, 16 * S + 2 - bos)
        rng_state = None
        seed = 0
        if exp_noise is None and self.sampling == "torch_cpu":
            rng_state = torch.get_rng_state()
            exp_noise = torch.stack([torch.empty(1, NUM_AUDIO_TOKENS + 1).exponential_(1)[0] for _ in range(n_max)])
        elif exp_noise is None:
            seed = int(torch.randint(0, 2**62, (1,)))
        eng.ar_decode(top_k=top_k, temperature=temperature, exp_noise=exp_noise, seed=seed, max_new_tokens=max_new_tokens)
        tokens, reason, n_pass = eng.ar_result()
        if rng_state is not None:
            # leave the global generator where the reference would: one draw per executed pass, including
            # the pass that trips the stop rule (valle.py:1040-1055)
            torch.set_rng_state(rng_state)
            for _ in range(tokens.numel() + 1):
                torch.empty(1, NUM_AUDIO_TOKENS + 1).exponential_(1)
        if tokens.numel() == 0 and max_new_tokens != 0:
            if not bos:
                raise SyntaxError("well trained model shouldn't reach here.")  # valle.py:1049-1052
        if self.print_eos:
            print(f"{self.MODEL_NAME} EOS [{P} -> {P + bos + tokens.numel()}]")  # valle.py:1054 / 646
        if Q == 1 or tokens.numel() == 0:
            codes = torch.zeros((tokens.numel(), Q), dtype=torch.int64)
            codes[:, 0] = tokens
            return codes.unsqueeze(0).to(self.device)

        text_nar = text
        if self.prefix_mode in [2, 4]:  # valle.py:1068-1079
            enrolled_len = int(enroll_x_lens.max().item())
            text_nar = torch.concat([text[:1], text[enrolled_len - 1:]])
        codes = eng.nar(text_nar, prompts, tokens, out_device=self.device)
        return codes.unsqueeze(0)


    @torch.no_grad()
    def inference_batch(self, utterances, top_k: int = -100, temperature: float = 1.0, seeds=None, batched_nar: bool = True,
                        batched_prefill: bool = True):
        """Engine extension (BASELINE configs[2]): ``utterances`` = list of (x, x_lens, y[, enroll_x_lens]) as for
        ``inference``; up to ``max_batch`` of them advance together, one shared weight stream per AR step, each with
        its own KV cache / sampler / stop rule; the NAR stages then run per utterance.  Returns a list of (1,T_i,Q)."""
        eng = self.engine()
        if eng.max_batch < 2:
            raise RuntimeError("construct the model with max_batch >= 2 for inference_batch")
        Q, bos = self.num_quantizers, int(self.ar_audio_prepend_bos)
        out = [None] * len(utterances)
        for g0 in range(0, len(utterances), eng.max_batch):
            group = utterances[g0 : g0 + eng.max_batch]
            for b, u in enumerate(group):
                x, x_lens, y = u[0], u[1], u[2]
                assert x.ndim == 2 and x_lens.ndim == 1 and y.ndim == 3 and y.shape[0] == 1 and torch.all(x_lens > 0)
                if x.shape[1] != int(x_lens.max()) or x.shape[0] != 1:
                    raise RuntimeError("x must be one unpadded sequence per utterance")
                if _ids_out_of_range(x, y[..., :Q]):
                    raise IndexError("index out of range in self")
                if not (batched_prefill and eng.mfma_rows):
                    eng.batch_prefill(b, x[0], y[0, :, 0].contiguous())
            if batched_prefill and eng.mfma_rows:  # one pass over the concatenated rows of the whole group
                eng.batch_prefill_all([u[0][0] for u in group], [u[2][0, :, 0].contiguous() for u in group])
            sd = [int(torch.randint(0, 2**62, (1,))) for _ in group] if seeds is None else list(seeds[g0 : g0 + len(group)])
            eng.batch_decode(len(group), top_k=top_k, temperature=temperature, seeds=sd)
            todo = []  # (index, text_nar, prompts, tokens) of the utterances that go through the NAR stages
            for b, u in enumerate(group):
                x, x_lens, y = u[0], u[1], u[2]
                enroll = u[3] if len(u) > 3 else None
                tokens, reason = eng.batch_result(b)
                if tokens.numel() == 0 and not bos:
                    raise SyntaxError("well trained model shouldn't reach here.")
                if Q == 1 or tokens.numel() == 0:
                    codes = torch.zeros((tokens.numel(), Q), dtype=torch.int64)
                    codes[:, 0] = tokens
                    out[g0 + b] = codes.unsqueeze(0).to(self.device)
                    continue
                text_nar = x[0]
                if self.prefix_mode in [2, 4]:
                    enrolled_len = int(enroll.max().item())
                    text_nar = torch.concat([x[0][:1], x[0][enrolled_len - 1:]])
                todo.append((g0 + b, text_nar, y[0, :, :Q].contiguous(), tokens))
            if todo and batched_nar:
                res = eng.nar_batch([t[1] for t in todo], [t[2] for t in todo], [t[3] for t in todo], out_device=self.device)
                for (i, *_), r in zip(todo, res):
                    out[i] = r.unsqueeze(0)
            else:
                for i, tn, pr, tk in todo:
                    out[i] = eng.nar(tn, pr, tk, out_device=self.device).unsqueeze(0)
        return out

    @torch.no_grad()
    def continual(self, x: torch.Tensor, x_lens: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """VALLE.continual (valle.py:1139-1238; reached by bin/infer.py:224-230 with --continual): the first half
        of y (at most 225 frames) is the prompt, codebook 0 of the rest is kept and its codebooks 1..7 are
        predicted by the NAR stages.  Returns (1, T - prefix_len, 8)."""
        assert x.ndim == 2, x.shape
        assert x_lens.ndim == 1, x_lens.shape
        assert y.ndim == 3, y.shape
        assert y.shape[0] == 1, y.shape
        assert torch.all(x_lens > 0)
        assert self.num_quantizers == 8
        if _ids_out_of_range(x, y):
            raise IndexError("index out of range in self")
        eng = self.engine()
        prefix_len = min(int(y.shape[1] * 0.5), 3 * 75)
        prompts = y[0, :prefix_len, :8].contiguous()
        rest0 = y[0, prefix_len:, 0].contiguous()
        codes = eng.nar(x[0], prompts, rest0, out_device=self.device, continual=True)
        return codes.unsqueeze(0)


class VALLF(VALLE):
    """The cross-attention variant (valle.py:49-719; ``--model-name VALL-F``): ``inference`` has VALLE.inference's signature and
    result (valle.py:566-710).  The text is embedded once as the memory of a TransformerDecoder whose target is the audio
    sequence alone; the reference masks memory positions >= x_lens (all-false for the unpadded batch-1 input it accepts).
    The reference's VALLF has no ``continual`` and no batched entry point; neither has this one."""

    MODEL_NAME = "VALL-F"

    def continual(self, *a, **k):
        raise AttributeError("'VALLF' object has no attribute 'continual'")  # valle.py:1139 defines it on VALLE only

    def inference_batch(self, *a, **k):
        raise NotImplementedError("VALL-F runs on the batch-1 path only")


def get_model(params) -> VALLE:
    """models/__init__.py:98-136 for --model-name VALL-E / VALL-F; the debug mel-Transformer is outside the hot path."""
    cfg = ModelConfig.from_params(params)
    if cfg.model_name.lower() in ("vall-f", "vallf"):
        cls = VALLF
    elif cfg.model_name.lower() in ("vall-e", "valle"):
        cls = VALLE
    else:
        raise NotImplementedError(f"model {cfg.model_name!r}: only VALL-E and VALL-F are built (DESIGN.md)")
    extra = {}
    get = params.get if isinstance(params, dict) else lambda k, d=None: getattr(params, k, d)
    for k in ("precision", "max_text", "max_audio", "sampling"):
        if get(k, None) is not None:
            extra[k] = get(k)
    return cls(cfg.decoder_dim, cfg.nhead, cfg.num_decoder_layers, norm_first=cfg.norm_first, add_prenet=cfg.add_prenet,
                 prefix_mode=cfg.prefix_mode, share_embedding=cfg.share_embedding, nar_scale_factor=cfg.scale_factor,
                 prepend_bos=cfg.prepend_bos, num_quantizers=cfg.num_quantizers, **extra)
