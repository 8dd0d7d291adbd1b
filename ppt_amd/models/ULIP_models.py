"""Mirror of models/ULIP_models.py for the hot path: LayerNorm, QuickGELU, ResidualAttentionBlock,
Transformer, PromptLearner, ULIP_WITH_IMAGE and the ULIP_PointBERT factory (ULIP_models.py:21-283,
443-512), with the reference's attribute names and state-dict keys so that main_cls.py-style
callers (`model(pc)`, `.prompt_learner`, `.point_encoder.blocks.blocks[-1]`, `.logit_scale`) and
ULIP checkpoints work unchanged.  Compute runs on libppt_hip.so through ppt_amd.engine.
"""
import contextlib
import json
import os
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from .. import blocks as _blk
from .. import engine, gradscale, graphs, ops

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
CONTEXT_LENGTH = 77


class LayerNorm(nn.LayerNorm):
    """ULIP_models.py:21-27: fp32 statistics whatever the input dtype.  On the device forward() is ppt_layernorm_fwd / _bwd."""

    def forward(self, x):
        if x.is_cuda:
            return _blk.layer_norm(x, self.weight, self.bias, self.eps)
        return super().forward(x.type(torch.float32)).type(x.dtype)


class QuickGELU(nn.Module):
    """ULIP_models.py:30-32."""

    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    """ULIP_models.py:35-56 (keys: attn.in_proj_weight/bias, attn.out_proj.*, ln_1, mlp.c_fc, mlp.c_proj, ln_2)."""

    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask

    def _is_causal(self, L):
        """The only mask the reference ever builds is the causal one (ULIP_models.py:224-230); the attention kernel applies
        it itself.  Any other additive mask is not supported by the kernels."""
        m = self.attn_mask
        if m is None:
            return False
        want = torch.full((L, L), float("-inf")).triu_(1)
        if m.shape[0] < L or not torch.equal(m[:L, :L].detach().float().cpu(), want):
            raise NotImplementedError("ResidualAttentionBlock: only attn_mask = None or the causal mask of build_attention_mask")
        return True

    def attention(self, x):
        """ULIP_models.py:49-51: nn.MultiheadAttention(x, x, x, attn_mask)[0]; x [L, N, D] (sequence first)."""
        L, N, D = x.shape
        a = self.attn
        heads = a.num_heads
        y = _blk.self_attention(x.transpose(0, 1), a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, heads,
                                float(D // heads) ** -0.5, self._is_causal(L), getattr(self, "precision", None))
        return y.transpose(0, 1)

    def forward(self, x):
        """ULIP_models.py:53-56; x [L, N, D]."""
        if not x.is_cuda:
            raise RuntimeError("ppt_amd modules run on the HIP device only (no CPU fallback)")
        x = x + self.attention(self.ln_1(x))
        h = self.ln_2(x)
        x = x + _blk.mlp(h, self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias,
                         ops.ACT_QUICKGELU, getattr(self, "precision", None))
        return x


class Transformer(nn.Module):
    """ULIP_models.py:59-67."""

    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width = width
        self.layers = layers
        self.heads = heads
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x):
        """ULIP_models.py:66-67; x [L, N, D]."""
        return self.resblocks(x)


# ---- prompt tokenisation (token-id table captured from the reference tokenizer)
_TOKENS = None


def _token_table():
    global _TOKENS
    if _TOKENS is None:
        with open(os.path.join(_DATA, "classnames.json")) as f:
            _TOKENS = json.load(f)
    return _TOKENS


def dataset_classnames(name):
    """class-name list of reference data/labels.json (utils/utils.py:118 loads it into args.classnames)."""
    return list(_token_table()["datasets"][name])


_BPE = {}


def _bpe(bpe_path=None):
    """The build's CLIP tokenizer (ppt_amd/tokenizer.py), or None when the merge table is not available."""
    from .. import tokenizer as T
    path = T.find_vocab(bpe_path)
    if path is None:
        return None
    if path not in _BPE:
        _BPE[path] = T.SimpleTokenizer(path)
    return _BPE[path]


def name_token_ids(name, bpe_path=None):
    """BPE ids of one class name / template word: the committed fixture for the reference's datasets, the tokenizer
     for anything else."""
    tab = _token_table()
    key = name.replace("_", " ")
    if key in tab["name_tokens"]:
        return list(tab["name_tokens"][key])
    tok = _bpe(bpe_path)
    if tok is None:
        raise KeyError(f"class name {key!r} has no captured token ids (ppt_amd/data/classnames.json) and the CLIP merge "
                       "table was not found (ppt_amd/tokenizer.py: PPT_BPE_VOCAB or ./utils/bpe_simple_vocab_16e6.txt.gz)")
    return tok.encode(key)


def tokenize_prompts(classnames, n_ctx, template_tokens=None, bpe_path=None, template_text=None):
    """Token ids of "X X ... X <name>." per class (ULIP_models.py:87-100) -> (ids [C,77] int64, name_lengths).
    Names of the reference's datasets come from the committed id table (each is known to tokenize to
    [sot] + ctx + name + [period, eot]).  Any other name goes through the tokenizer the way the reference does it: the
    WHOLE string "<prefix> <name>." is encoded at once (a name ending in punctuation merges with the period into one
    BPE piece), and name_lengths counts the pieces of the name encoded alone."""
    tab = _token_table()
    ids, lens = [], []
    for name in classnames:
        key = name.replace("_", " ")
        ctx = template_tokens if template_tokens is not None else [tab["placeholder"]] * n_ctx
        if key in tab["name_tokens"]:
            nt = list(tab["name_tokens"][key])
            row = [tab["sot"]] + list(ctx) + nt + [tab["period"], tab["eot"]]
            n_name = len(nt)
        else:
            tok = _bpe(bpe_path)
            if tok is None:
                name_token_ids(name, bpe_path)                  # raises the KeyError that explains what is missing
            prefix = template_text if template_text is not None else " ".join(["X"] * n_ctx)
            row = [tab["sot"]] + tok.encode(prefix + " " + key + ".") + [tab["eot"]]
            n_name = len(tok.encode(key))
        if len(row) > CONTEXT_LENGTH:
            raise RuntimeError(f"prompt for {name!r} exceeds the context length {CONTEXT_LENGTH}")
        ids.append(row + [0] * (CONTEXT_LENGTH - len(row)))
        lens.append(n_name)
    return torch.tensor(ids, dtype=torch.long), lens


class PromptLearner(nn.Module):
    """ULIP_models.py:70-151.  The only parameter is `learnable_tokens` [n_ctx, width]."""

    def __init__(self, token_embeding, kwargs):
        super().__init__()
        self.class_name_position = kwargs.class_name_position
        self.classnames = kwargs.classnames
        self.transformer_width = kwargs.transformer_width
        self.device = kwargs.device
        template_tokens = template_text = None
        if kwargs.template_init != '':
            template_text = kwargs.template_init.replace("_", " ")
            words = template_text.split(' ')
            template_tokens = []
            for wd in words:
                template_tokens += name_token_ids(wd)
            if len(template_tokens) != len(words):       # ULIP_models.py:82: one learnable token per template WORD
                raise RuntimeError("every word of template_init must be a single BPE token "
                                   f"({kwargs.template_init!r} -> {len(template_tokens)} tokens for {len(words)} words)")
            self.num_learnable_prompt_tokens = len(words)
        else:
            self.num_learnable_prompt_tokens = kwargs.num_learnable_prompt_tokens
        self.learnable_tokens = nn.Parameter(torch.empty(self.num_learnable_prompt_tokens, self.transformer_width))
        self.tokenized_prompts, self.name_lengths = tokenize_prompts(self.classnames, self.num_learnable_prompt_tokens,
                                                                     template_tokens, template_text=template_text)
        # the frozen prefix / class-name / suffix embeddings are looked up ONCE at
        # construction (before any weight is initialised or loaded) and are neither parameter nor buffer.
        with torch.no_grad():
            self.embedding = token_embeding(self.tokenized_prompts).detach().clone()
        self._index = None

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.embedding = fn(self.embedding)
        self._index = None
        return self

    def _scatter_index(self, device):
        """(class, position, token) triplets of every learnable-token slot (ULIP_models.py:112-148)."""
        if self._index is None or self._index[0].device != device:
            n = self.num_learnable_prompt_tokens
            C = len(self.classnames)
            pos = torch.empty((C, n), dtype=torch.long)
            for i in range(C):
                L = self.name_lengths[i]
                if self.class_name_position == "end":
                    pos[i] = 1 + torch.arange(n)
                elif self.class_name_position == "middle":
                    half = n // 2
                    pos[i, :half] = 1 + torch.arange(half)
                    pos[i, half:] = 1 + half + L + torch.arange(n - half)
                elif self.class_name_position == "front":
                    pos[i] = 1 + L + torch.arange(n)
                else:
                    raise ValueError(f'`class_name_position`: {self.class_name_position} not in supported modes '
                                     f'["front", "middle", "end"]')
            src = torch.empty((C, CONTEXT_LENGTH), dtype=torch.long)     # where every output slot comes from
            for i in range(C):
                L = self.name_lengths[i]
                order = list(range(CONTEXT_LENGTH))
                if self.class_name_position == "middle":
                    half = n // 2
                    # [prefix | ctx[:half] | class | ctx[half:] | rest]; the cached embedding is [prefix | ctx | class | rest]
                    order = [0] + list(range(1, 1 + half)) + list(range(1 + n, 1 + n + L)) \
                        + list(range(1 + half, 1 + n)) + list(range(1 + n + L, CONTEXT_LENGTH))
                elif self.class_name_position == "front":
                    order = [0] + list(range(1 + n, 1 + n + L)) + list(range(1, 1 + n)) + list(range(1 + n + L, CONTEXT_LENGTH))
                src[i] = torch.tensor(order)
            cls = torch.arange(C).view(C, 1).expand(C, n)
            self._index = (cls.to(device), pos.to(device), src.to(device))
        return self._index

    def forward(self):
        """-> prompts [C,77,width]: frozen embeddings with the learnable tokens spliced in."""
        if self.class_name_position not in ("front", "middle", "end"):
            raise ValueError(f'`class_name_position`: {self.class_name_position} not in supported modes '
                             f'["front", "middle", "end"]')
        dev = self.learnable_tokens.device
        cls, pos, src = self._scatter_index(dev)
        emb = self.embedding.to(dev)
        base = torch.gather(emb, 1, src.unsqueeze(-1).expand(-1, -1, emb.shape[-1]))
        return base.index_put((cls, pos), self.learnable_tokens.unsqueeze(0).expand(cls.shape[0], -1, -1))

    def eot_positions(self):
        return self.tokenized_prompts.argmax(dim=-1)

    def row_layout(self, positional, L, P):
        """Constants of the text tower's input in its row layout for ops.prompt_rows / prompt_rows_bwd:
        base [M, W] = frozen embedding + positional embedding per row, slot [M] i32 = index of the learnable token that
        overwrites the row (-1: none), pos_rows [M, W] = positional embedding per row, rows_of [n_tok, C] i32 = the rows each
        learnable token appears in (ascending, -1 padded), M, eot_rows [C] i64 = the row of every class's EOT token.
        Layout (engine.text_tower_forward): P shared rows + C (L - P) own rows, or C L rows when P == 0.
        Cached per (embedding, positional embedding, L, P)."""
        emb = self.embedding
        key = (emb.data_ptr(), emb._version, positional.data_ptr(), positional._version, L, P, self.class_name_position)
        cache = self.__dict__.setdefault("_layout_caches", {})
        if key not in cache:
            dev = self.learnable_tokens.device
            cls, pos, src = self._scatter_index(torch.device("cpu"))
            C, n = pos.shape
            tok_of = torch.full((C, CONTEXT_LENGTH), -1, dtype=torch.int32)
            tok_of[cls, pos] = torch.arange(n, dtype=torch.int32).view(1, n).expand(C, n)
            eot = self.tokenized_prompts.argmax(dim=-1).tolist()
            rows, eot_rows = [], [0] * C                 # rows: (class, position) or None for an unused row
            if P:
                rows += [(0, q) for q in range(P)]
                for c in range(C):
                    eot_rows[c] = len(rows) + (eot[c] - P)
                    rows += [(c, q) for q in range(P, L)]
            else:
                for c in range(C):
                    eot_rows[c] = len(rows) + eot[c]
                    rows += [(c, q) for q in range(L)]
            used = torch.tensor([r is not None for r in rows])
            rc = torch.tensor([r[0] if r is not None else 0 for r in rows]), torch.tensor([r[1] if r is not None else 0 for r in rows])
            slot = torch.where(used, tok_of[rc[0], rc[1]], torch.full((len(rows),), -1, dtype=torch.int32)).contiguous()
            with torch.no_grad():
                frozen = emb.detach().to("cpu").float()
                pos_rows = (positional.detach().to("cpu").float()[rc[1]] * used.view(-1, 1)).contiguous()
                base = (frozen[rc[0], src[rc[0], rc[1]]] * used.view(-1, 1) + pos_rows).contiguous()
            rows_of = torch.full((n, C), -1, dtype=torch.int32)
            fill = [0] * n
            for i, t in enumerate(slot.tolist()):
                if t >= 0:
                    rows_of[t, fill[t]] = i
                    fill[t] += 1
            if len(cache) > 8:
                cache.clear()
            cache[key] = (base.to(dev), slot.to(dev), pos_rows.to(dev), rows_of.to(dev), len(rows),
                          torch.tensor(eot_rows, dtype=torch.long).to(dev))
        return cache[key]

    def shared_prefix(self):
        """Number of leading positions whose prompt rows are the same for EVERY class: the start token (one token id, hence
        one embedding row: ULIP_models.py:102 looks the prompts up in token_embedding) followed by the learnable context tokens
        that come before the class name (ULIP_models.py:112-148: all 32 for "end", the first half for "middle", none for
        "front").  Checked against the cached embedding itself (a caller may have replaced it with per-class rows: then 0)."""
        emb = self.embedding
        key = (emb.data_ptr(), str(emb.device), emb._version, self.class_name_position)
        if getattr(self, "_prefix_cache", (None, 0))[0] != key:
            n = self.num_learnable_prompt_tokens
            cand = {"middle": 1 + n // 2, "end": 1 + n}.get(self.class_name_position, 1)
            same = bool((self.tokenized_prompts[:, 0] == self.tokenized_prompts[0, 0]).all()) and \
                bool((emb[:, 0] == emb[0:1, 0]).all().item())                       # (one host read per embedding, cached)
            self._prefix_cache = (key, cand if (same and cand >= 8 and len(self.classnames) > 1) else 0)
        return self._prefix_cache[1]


class _TextTowerFn(torch.autograd.Function):
    """encode_text as one autograd node.  After graphs.WARMUP_CALLS eager calls the forward (and the input-gradient
    backward) of a given shape are replayed from a captured hipGraph (ppt_amd/graphs.py)."""

    @staticmethod
    def forward(ctx, model, prompts, trusted_prefix=False):
        sd, cache = model._live_state(), model._cache()
        save = bool(ctx.needs_input_grad[1])
        prompts = prompts.contiguous().float()
        eot = model._eot(prompts.device)
        heads, layers = model.transformer.heads, model.transformer.layers
        eff = model._text_len() if model.truncate_text_to_eot else None
        # prefix sharing is a property of the PROMPTS: PromptLearner's own output has it by construction (trusted_prefix, from
        # _text_raw); any other tensor handed to the public encode_text is checked (ULIP_WITH_IMAGE._prefix_of) and falls back
        # to the general evaluation when its leading positions differ between classes
        pre = model.prompt_learner.shared_prefix() if model.share_text_prefix else 0
        if pre and not trusted_prefix:
            pre = model._prefix_of(prompts, pre)

        prio = model.chain_priority()

        def run(pr):
            with ops.wave_priority(prio):
                return engine.text_tower_forward(sd, cache, pr, eot, heads, layers, save, eff_len=eff, prefix=pre)

        key = ("text_fwd", tuple(prompts.shape), save, eff, pre, cache.dtype, prio)
        gc = model._graphs
        ctx.model, ctx.graph = model, None
        ctx.grad_scale = gradscale.current(cache.dtype) if save else 1.0       # the backward's power-of-two scale, fixed now
        if prompts.is_cuda and model.use_hip_graphs and ops.profiler is None and gc.ready(key):
            def build():
                def fn(pr):
                    out, saved = run(pr)
                    return (out,), saved
                return graphs.GraphedCall(fn, [prompts])
            g = gc.get(key, build)
            (out,), saved = g(prompts)
            g.generation = getattr(g, "generation", 0) + 1
            ctx.graph, ctx.key, ctx.generation = g, key, g.generation
            if not model._handoff:
                out = out.clone()
        else:
            out, saved = run(prompts)
        ctx.saved = saved
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.model
        sd, cache = m._live_state(), m._cache()
        dout = dout.float()
        if dout.is_cuda:
            dout.record_stream(torch.cuda.current_stream())     # produced on the caller's stream, read on the text stream
        prio = m.chain_priority()
        S = ctx.grad_scale
        if ctx.graph is None:
            with ops.wave_priority(prio):
                return None, engine.text_tower_backward(sd, cache, ctx.saved, dout, grad_scale=S), None
        if ctx.graph.generation != ctx.generation:
            raise RuntimeError("the text tower's captured activations were overwritten by a later forward; set "
                               "model.use_hip_graphs = False to keep several forwards alive before backward")
        saved, fwd = ctx.saved, ctx.graph

        def build():
            def fn(d):
                with ops.wave_priority(prio):
                    return (engine.text_tower_backward(sd, cache, saved, d, grad_scale=S),), None
            return graphs.GraphedCall(fn, [dout.contiguous()], pool=fwd.pool())
        (dp,), _ = m._graphs.get(("text_bwd",) + ctx.key[1:] + (S,), build)(dout)
        return None, dp.clone(), None


class _TextTowerTokensFn(torch.autograd.Function):
    """text features as a function of PromptLearner.learnable_tokens directly: the splice (ULIP_models.py:104-151), the
    positional add (:210) and the row layout of the tower in ONE kernel (ppt_prompt_rows) instead of gather + index_put +
    cat + the add; the backward folds the tower's input gradient onto the tokens in one kernel (ppt_prompt_rows_bwd)
    instead of zeros + slice copies + index_put's backward.  Same graphs / saved activations as _TextTowerFn."""

    @staticmethod
    def forward(ctx, model, tokens):
        sd, cache = model._live_state(), model._cache()
        save = bool(ctx.needs_input_grad[1])
        pl = model.prompt_learner
        eot = model._eot(tokens.device)
        heads, layers = model.transformer.heads, model.transformer.layers
        C, Lfull = len(pl.classnames), model.context_length
        eff = model._text_len() if model.truncate_text_to_eot else Lfull
        pre = pl.shared_prefix() if model.share_text_prefix else 0
        if not (0 < pre < eff and C > 1):
            pre = 0
        base, slot, pos_rows, rows_of, M, eot_rows = pl.row_layout(sd["positional_embedding"], eff, pre)
        tok = tokens.detach().float().contiguous()

        prio = model.chain_priority()

        def run(tk):
            with ops.wave_priority(prio):
                x0 = ops.prompt_rows(base, slot, tk, pos_rows)
                return engine.text_tower_forward(sd, cache, None, eot, heads, layers, save, eff_len=eff, prefix=pre, rows_in=(x0, C, Lfull))

        key = ("text_fwd_tok", tuple(tok.shape), save, eff, pre, cache.dtype, prio)
        gc = model._graphs
        ctx.model, ctx.graph, ctx.rows_of, ctx.n_tok = model, None, rows_of, tok.shape[0]
        ctx.grad_scale = gradscale.current(cache.dtype) if save else 1.0       # the backward's power-of-two scale, fixed now
        if tok.is_cuda and model.use_hip_graphs and ops.profiler is None and gc.ready(key):
            def build():
                def fn(tk):
                    out, saved = run(tk)
                    return (out,), saved
                return graphs.GraphedCall(fn, [tok])
            g = gc.get(key, build)
            (out,), saved = g(tok)
            g.generation = getattr(g, "generation", 0) + 1
            ctx.graph, ctx.key, ctx.generation = g, key, g.generation
            if not model._handoff:                       # (forward_loss copies it into the head graph's input at once)
                out = out.clone()
        else:
            out, saved = run(tok)
        ctx.saved = saved
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.model
        sd, cache = m._live_state(), m._cache()
        dout = dout.float()
        if dout.is_cuda:
            dout.record_stream(torch.cuda.current_stream())
        rows_of, n_tok = ctx.rows_of, ctx.n_tok

        prio = m.chain_priority()
        S = ctx.grad_scale

        def run(d, saved):
            # the scale enters in the first product of the tower's backward and leaves in the kernel that folds the rows onto
            # the tokens: no launch more than without it
            with ops.wave_priority(prio):
                return ops.prompt_rows_bwd(engine.text_tower_backward(sd, cache, saved, d, grad_scale=S), rows_of, n_tok, scale=1.0 / S)

        if ctx.graph is None:
            return None, run(dout.contiguous(), ctx.saved)
        if ctx.graph.generation != ctx.generation:
            raise RuntimeError("the text tower's captured activations were overwritten by a later forward; set "
                               "model.use_hip_graphs = False to keep several forwards alive before backward")
        saved, fwd = ctx.saved, ctx.graph

        def build():
            def fn(d):
                return (run(d, saved),), None
            return graphs.GraphedCall(fn, [dout.contiguous()], pool=fwd.pool())
        (dt,), _ = m._graphs.get(("text_bwd_tok",) + ctx.key[1:] + (S,), build)(dout)
        # learnable_tokens is a leaf whose .grad exists (a view of the Trainer's flat buffer): autograd ADDS dt into it on this
        # stream before anything can replay the graph again -- no copy needed then
        tok_grad = m.prompt_learner.learnable_tokens.grad
        return None, (dt if (m._handoff_grad and tok_grad is not None) else dt.clone())


class _MatmulNT(torch.autograd.Function):
    """a [M,K] @ b[N,K]^T with fp32 results.  prec = torch.float32: fp32 operands on the fp32 MFMA (the classification
    heads: projection and logits of a few dozen rows); prec = torch.float16 / torch.bfloat16: 16-bit operands, fp32
    accumulation -- the per-POINT head of part segmentation (32 768 rows x 50 parts) in the performance mode, where the fp32
    MFMA (1/16 of the 16-bit rate) cost 1.06 ms of a 14 ms step.  The 16-bit backward is a gradient-scaled stage of its own
    (ppt_amd/gradscale.py): dC is multiplied by the power of two S while it is converted, dA leaves through the GEMM's row scale
    1 / S, dB by one small multiply -- d logits of a mean over 32 768 rows is ~1e-5 per element, a subnormal in half."""

    @staticmethod
    def forward(ctx, a, b, prec):
        a, b = a.contiguous().float(), b.contiguous().float()
        ctx.save_for_backward(a, b)
        ctx.prec = prec if (prec in ops.HALF and a.shape[1] % 8 == 0) else torch.float32
        ctx.grad_scale = gradscale.current(ctx.prec, default_rows=a.shape[0])
        ctx.inv = gradscale.inv_tensor(ctx.grad_scale, a.device) if ctx.grad_scale != 1.0 else None
        if ctx.prec in ops.HALF:
            return ops.gemm(ops.convert(a, prec), ops.convert(b, prec), out_dtype=torch.float32)
        return ops.gemm(a, b, out_dtype=torch.float32)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = dc.contiguous().float()
        da = db = None
        T = ctx.prec if ctx.prec in ops.HALF else torch.float32
        mult = 8 if T in ops.HALF else 4
        S = ctx.grad_scale
        rs = dict(row_scale=ctx.inv, row_scale_rows=dc.shape[0]) if S != 1.0 else {}
        if ctx.needs_input_grad[0]:                      # dA[M,K] = dC[M,N] @ B[N,K]; N padded to the chunk size
            n = dc.shape[1]
            dcp = dc if n % mult == 0 else torch.nn.functional.pad(dc, (0, mult - n % mult))
            da = ops.gemm(ops.convert(dcp, T, scale=S), ops.transpose(ops.convert(b, T), pad_to=mult), out_dtype=torch.float32, **rs)
        if ctx.needs_input_grad[1]:                      # dB[N,K] = dC[M,N]^T @ A[M,K]
            db = ops.gemm_tn_splitk(ops.convert(dc, T, scale=S), ops.convert(a, T))
            if S != 1.0:
                db = db * (1.0 / S)
        return da, db, None


HEAD_ON_TEXT_STREAM = os.environ.get("PPT_HEAD_ON_TEXT_STREAM", "1") != "0"


class _HeadLossFn(torch.autograd.Function):
    """(loss, logits) of one batch for the frozen-point-side case; the gradient w.r.t. the text features is formed in
    the forward (engine.head_loss_forward_backward) and handed out by backward.  Replayed from a hipGraph after
    graphs.WARMUP_CALLS eager calls."""

    @staticmethod
    def forward(ctx, model, pc_feat, text_raw, labels, smoothing):
        wt = engine._f32_cache(model._cache()).get(model.pc_projection, "wt")
        scale = model.logit_scale.detach()
        proj = model.pc_projection.detach()
        feat, traw, lab = pc_feat.detach().float().contiguous(), text_raw.detach().float().contiguous(), labels.contiguous()

        prio = model.chain_priority()

        def run(f, t, l):
            with ops.wave_priority(prio):
                return engine.head_loss_forward_backward(f, wt, t, scale, l, smoothing, w=proj), None

        key = ("head", tuple(feat.shape), tuple(traw.shape), float(smoothing), prio)
        gc = model._graphs
        if feat.is_cuda and model.use_hip_graphs and ops.profiler is None and gc.ready(key):
            g = gc.get(key, lambda: graphs.GraphedCall(run, [feat, traw, lab]))
            (loss, logits, d_raw), _ = g(feat, traw, lab)
            g.generation = getattr(g, "generation", 0) + 1
            ctx.graph, ctx.generation = g, g.generation
            loss, logits = loss.clone(), logits.clone()  # (d_raw stays in the graph's buffer: read by backward before the next replay)
        else:
            (loss, logits, d_raw), _ = run(feat, traw, lab)
            ctx.graph = None
        ctx.d_raw = d_raw
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, dloss, _dlogits):
        if ctx.graph is not None and ctx.graph.generation != ctx.generation:
            raise RuntimeError("the head's captured text-feature gradient was overwritten by a later forward_loss; set "
                               "model.use_hip_graphs = False to keep several forwards alive before backward")
        return None, None, ctx.d_raw * dloss, None, None


def matmul_nt(a, b, prec=torch.float32):
    return _MatmulNT.apply(a, b, prec)


class ULIP_WITH_IMAGE(nn.Module):
    """ULIP_models.py:154-283 (text transformer + PromptLearner + point encoder + projections)."""

    def __init__(self, point_encoder, **kwargs):
        super().__init__()
        kwargs = SimpleNamespace(**kwargs)
        self.task = kwargs.task
        self.context_length = kwargs.context_length
        self.device = kwargs.device
        self.transformer = Transformer(width=kwargs.transformer_width, layers=kwargs.transformer_layers,
                                       heads=kwargs.transformer_heads, attn_mask=self.build_attention_mask())
        self.vocab_size = kwargs.vocab_size
        self.token_embedding = nn.Embedding(kwargs.vocab_size, kwargs.transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(self.context_length, kwargs.transformer_width))
        self.ln_final = LayerNorm(kwargs.transformer_width)
        self.text_projection = nn.Parameter(torch.empty(kwargs.transformer_width, kwargs.embed_dim))
        self.pc_projection = nn.Parameter(torch.empty(kwargs.pc_feat_dims, kwargs.embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.point_encoder = point_encoder
        self.prompt_learner = PromptLearner(self.token_embedding, kwargs)
        self.tokenized_prompts = self.prompt_learner.tokenized_prompts
        self.initialize_parameters()
        self._wc = None
        self._sd = None
        self._eot_pos = None
        self._text_stream = None
        self._te_cache = None
        self._text_len_cache = None
        self._graphs = graphs.GraphCache()
        self.use_hip_graphs = True          # replay the text tower from captured hipGraphs after two eager calls
        self.overlap_text_tower = True      # run the (input-independent) text tower on a side stream
        self.truncate_text_to_eot = True    # causal mask + EOT pooling: positions after the last EOT are dead work
        self.fused_prompt_rows = os.environ.get("PPT_FUSED_PROMPT_ROWS", "1") != "0"     # PromptLearner splice + pos add: one kernel
        # positions in front of the class name are the same in every prompt: computed once (PPT_SHARE_TEXT_PREFIX=0: A/B runs)
        self.share_text_prefix = os.environ.get("PPT_SHARE_TEXT_PREFIX", "1") != "0"
        self.text_f16 = os.environ.get("PPT_TEXT_F16", "1") != "0"
        # validate() (no_grad, eval mode): True = the caller vouches that every `pc` passed to forward() is COMPLETE in device
        # memory when forward() is called (a resident tensor, or a loader that synchronised its copy stream) -- the grouping /
        # tokenizer stage of a batch then runs on its own stream as soon as forward() is called, i.e. under the previous batch's
        # transformer blocks (as train.Trainer.inputs_ready does for training steps)
        self.eval_inputs_ready = False
        self._chain_prio = None
        self.health = None                  # ppt_amd.health.Monitor (train.Trainer installs one in the mixed 16-bit mode)
        # The text tower's operand format: None = what the mode says (IEEE half in "mixed16"); torch.float32 = fp32 operands whatever
        # the mode (PPT_TEXT_PRECISION=fp32, or calibrate_text_precision() below found half too coarse for THESE weights).
        self.text_precision = torch.float32 if os.environ.get("PPT_TEXT_PRECISION", "").lower() in ("fp32", "float32") else None
        # a text tower on fp32 operands INSIDE the mixed mode (the override above, or the load-time check's verdict) forms its
        # products from hi + lo half pairs (split16) instead of on the fp32 MFMA; PPT_TEXT_SPLIT16=0: the fp32 MFMA
        self.text_split16 = os.environ.get("PPT_TEXT_SPLIT16", "1") != "0"
        self.split16 = False
        # split16's operand pre-scales (2^a for activations / gradients, 2^b for weights) are THIS model's: b follows its weight range
        # (_fit_split16_range, re-run per weight set); handed to ops with the flag by every _cache() call
        self._split_pow2_default = tuple(ops.SPLIT16_POW2)
        self.split_pow2 = self._split_pow2_default
        self._split_fit_pending = True
        self.text_calibration = None        # what calibrate_text_precision measured: {"rel_l2": ..., "threshold": ..., "demoted": bool}
        self._text_calibrated = False
        # stages this model's monitor moved from IEEE half to bf16 (health.demote); ONE set per model, shared with the point
        # encoder and every WeightCache of either
        self.demoted = point_encoder.__dict__.setdefault("demoted", set()) if isinstance(point_encoder, nn.Module) else set()
        self._handoff = False               # inside forward_loss: graph outputs go straight into the next graph's input (no clone)
        self._handoff_grad = os.environ.get("PPT_HANDOFF", "1") != "0"

    # ---- reference helpers ------------------------------------------------------------------
    def build_attention_mask(self):
        """ULIP_models.py:224-230 (kept for API parity; the attention kernel applies the causal mask itself)."""
        mask = torch.empty(self.context_length, self.context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask

    def initialize_parameters(self):
        """ULIP_models.py:232-248."""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.prompt_learner.learnable_tokens, std=0.02)
        proj_std = (self.transformer.width ** -0.5) * ((2 * self.transformer.layers) ** -0.5)
        attn_std = self.transformer.width ** -0.5
        fc_std = (2 * self.transformer.width) ** -0.5
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=self.transformer.width ** -0.5)
        nn.init.normal_(self.pc_projection, std=512 ** -0.5)

    # ---- plumbing -----------------------------------------------------------------------------
    @property
    def precision(self):
        return getattr(self.point_encoder, "precision", torch.bfloat16)

    MIXED16 = "mixed16"

    @property
    def precision_name(self):
        """"mixed16" (performance mode) or "fp32" (parity mode)."""
        return ("split16" if getattr(self, "split16", False) else "fp32") if self.precision == torch.float32 else self.MIXED16

    def set_precision(self, mode):
        """"mixed16" -- the performance mode: 16-bit MFMA operands with fp32 accumulation, residual streams / statistics /
        gradients in fp32; the operand FORMAT is chosen per stage (IEEE half for the CLIP text tower, the PointBERT tokenizer and
        blocks, the part-seg decoder and per-point head; bf16 for PointNet++ / PointMLP: engine.*_F16) -- or
        "fp32" / torch.float32 -- the parity mode (fp32 operands on the fp32 MFMA) -- or "split16": the parity mode's fp32 storage,
        statistics and VALU attention, with every GEMM product formed on the 16-bit matrix pipe from hi + lo IEEE-half pairs of the
        fp32 operands (ppt_gemm_params.split16: 22 significand bits per operand, ~2x the fp32 MFMA's rate, results at least as
        close to an fp64 product as the fp32 MFMA's; gradient stages are scaled as in mixed16, ppt_amd/gradscale.py).
        `torch.bfloat16` is accepted as a DEPRECATED alias of "mixed16" (rounds 1-2 ran that mode in bf16 throughout and named it
        after the format; it has not been a statement about the operand format since round 3)."""
        split16 = False
        if isinstance(mode, str):
            key = mode.lower()
            if key not in (self.MIXED16, "fp32", "float32", "split16"):
                raise ValueError(f'set_precision: expected "mixed16", "split16" or "fp32", got {mode!r}')
            dtype = torch.bfloat16 if key == self.MIXED16 else torch.float32
            split16 = key == "split16"
        elif mode is torch.bfloat16 or mode is torch.float16:
            import warnings
            warnings.warn('set_precision(torch.bfloat16 / torch.float16) selects the MIXED 16-bit mode (IEEE half operands in the '
                          'normalised stages, bf16 in PointNet++ / PointMLP), not a format: write set_precision("mixed16")',
                          DeprecationWarning, stacklevel=2)
            dtype = torch.bfloat16
        elif mode is torch.float32:
            dtype = torch.float32
        else:
            raise ValueError(f'set_precision: expected "mixed16", "fp32" or torch.float32, got {mode!r}')
        # (internally the mode marker stays a dtype: torch.bfloat16 = mixed16, torch.float32 = parity; split16 is the parity mode's
        # storage with every ppt_gemm product formed from hi + lo half pairs -- ops.set_split16, said by every _cache() call)
        self.split16 = split16
        for m in self.modules():
            if m is not self and (hasattr(m, "_cache") or hasattr(m, "_wc")):
                m.split16 = split16
        if split16:
            self._fit_split16_range()
        self.point_encoder.precision = dtype
        self.point_encoder._wc = None
        self._text_calibrated = False
        if self.text_calibration is not None and self.text_calibration.get("demoted"):
            self.text_precision = None              # (a calibration's verdict belongs to the mode and weights it was taken on)
        self.text_calibration = None
        self._graphs.clear()
        if hasattr(self.point_encoder, "_graphs"):
            self.point_encoder._graphs.clear()
        if hasattr(self.point_encoder, "encoder"):
            self.point_encoder.encoder.precision = dtype
        for m in self.modules():                     # the callable sub-modules (ppt_amd/blocks.py) follow the model's mode
            if type(m).__name__ in ("Mlp", "Attention", "Block", "ResidualAttentionBlock"):
                m.precision = dtype
        return self

    def _fit_split16_range(self):
        """split16 multiplies every weight by 2^b (default b = 4, ops.SPLIT16_POW2 at import) before splitting it into hi + lo
        halves, and half stops at 65 504: with weights (or BatchNorm-folded weights) beyond ~2 000 the hi halves would saturate.
        Checked per WEIGHT SET -- set_precision("split16"), and again at the first _cache() after load_state_dict / reset_caches
        (a fit at set_precision alone never saw a checkpoint loaded afterwards) -- on the matrices this model holds:
        b is lowered (with a warning) until 4 x max|w| x 2^b stays below half's range; the factor 4 is room for the BatchNorm folds
        (w x gamma / sigma).  The result is THIS model's (`split_pow2`, handed to ops.set_split16 with the flag by every _cache()
        call), so one model's fit no longer changes another's."""
        self._split_fit_pending = False
        mats = [q.detach() for q in self.parameters() if q.dim() >= 2 and q.is_floating_point()]
        a, b = self._split_pow2_default
        if mats:
            mx = max(float(q.abs().max()) for q in mats)
            b_fit = b
            while b_fit > -8 and 4.0 * mx * 2.0 ** b_fit >= 32768.0:
                b_fit -= 1
            if b_fit != b:
                import warnings
                warnings.warn(f"split16: max |weight| = {mx:.4g}; the weight operand's pre-scale is lowered from 2^{b} to 2^{b_fit} "
                              f"(this model's split_pow2) to keep the hi halves inside IEEE half's range")
            b = b_fit
        if (a, b) != self.split_pow2:
            self.split_pow2 = (a, b)
            self._graphs.clear()                    # (a captured launch carries the pre-scales as kernel arguments)
            if hasattr(self.point_encoder, "_graphs"):
                self.point_encoder._graphs.clear()

    def _cache(self):
        # The text tower's operand format in the performance mode is IEEE half, not bf16 (PPT_TEXT_F16=0: bf16): same MFMA rate,
        # 11 significand bits instead of 8.  tools/bf16_error.py attributes 69 % of the bf16 mode's squared logits error and 89 %
        # of its token-gradient error to the bf16 text tower -- almost all of it to the attention half of its layers, where the
        # rounding of q and k is amplified by the softmax.  Its activations are LayerNorm outputs, projections of them and
        # softmax weights (the residual stream is fp32), far inside fp16's range; gradients are O(1e-2 .. 1).
        want = getattr(self, "text_precision", None)                        # (explicit override: error-budget experiments)
        if want is None:
            want = torch.float16 if (self.precision == torch.bfloat16 and self.text_f16) else self.precision
        if self._wc is None or self._wc.dtype != want:
            self._wc = engine.WeightCache(want, self.demoted)
        # split16: the whole model's mode, or the text tower alone when its load-time check sent it to fp32 operands
        on = self.split16 or (want == torch.float32 and self.precision != torch.float32 and self.text_split16)
        if on and self._split_fit_pending and not torch.cuda.is_current_stream_capturing():
            self._fit_split16_range()               # (new weights since the last fit: load_state_dict / reset_caches)
        ops.set_split16(on, self.split_pow2)
        return self._wc

    def _live_state(self):
        ref = self.positional_embedding
        if self._sd is None or self._sd[1] is not ref or self._sd[2] != ref.device:
            sd = {k: v for k, v in self.state_dict(keep_vars=True).items() if not k.startswith("point_encoder.")}
            self._sd = (sd, ref, ref.device)
        return self._sd[0]

    def chain_priority(self):
        """1 when the prompt chain is the step's critical path -- nothing but the PromptLearner trains, so the point tower
        never waits for the optimizer -- else 0 (a training point side makes the tower critical: measured +0.2-0.5 % step
        time with the chain prioritised on C3 / C5).  PPT_CHAIN_PRIO=0|1 overrides.  See csrc/ppt_common.h PPT_PRIO."""
        if self._chain_prio is None:
            env = os.environ.get("PPT_CHAIN_PRIO")
            if env in ("0", "1"):
                self._chain_prio = int(env)
            else:
                # (a frozen encoder whose tower is much longer than the chain -- PointMLP: 4.6 ms -- is the critical path itself:
                # 4.655 -> 4.686 ms with the chain prioritised; such encoders carry chain_priority_hint = 0)
                hint = getattr(self.point_encoder, "chain_priority_hint", 1)
                self._chain_prio = int(bool(hint) and all(n.startswith("prompt_learner.") or not q.requires_grad
                                                          for n, q in self.named_parameters()))
        return self._chain_prio

    def _tower_room(self):
        """Context for the point tower's launches of a TRAINING step whose critical path is the prompt chain (chain_priority):
        its persistent kernels leave CUs to the text stream (ops.persistent_occupancy; PPT_TOWER_OCCUPANCY percent.  Round 2: 60,
        the optimum of 200 / 150 / 100 / 75 / 60 / 50 / 40 / 30 % of a workgroup per CU, C2 3.67 -> 3.50 ms.  Round 3, with the
        prompt chain's GEMMs 6-23 % faster the chain needs less room: 60 / 70 / 80 / 85 / 90 / 100 % -> 3.344 / 3.314 / 3.272 /
        3.281 / 3.400 / 3.613 ms on one box with the tokenizer in the tower's graph; with the tokenizer running ahead on its own
        stream (PointTransformer.tokenize_ahead, the default) 50 / 60 / 70 % -> 3.249 / 3.212 / 3.205 ms and 80 % 3.27: default 70.
        The step sits on a plateau now -- 3.20 ... 3.27 ms for every combination of {grouping ahead, tokenizer ahead, neither} x
        {60 ... 85 %} on that box: what remains is the kernels' own resource time.  Round 4, re-measured on two boxes after the
        prompt chain lost its un-scaling multiply and gained the one-launch optimizer: 60 / 70 / 80 / 85 / 90 / 100 % -> 3.309 /
        3.233 / 3.154 | 3.115 / 3.125 / 3.197 / 3.581 ms IN ORDER (tools/step_parts.py), but with the grouping / tokenizer stage
        running ahead (Trainer.inputs_ready, bench.py's loop) the chain is the longer side again and wants more room: ahead 70 /
        80 % -> 3.064 / 3.155 ms, in order 70 / 80 / 90 % -> 3.185 / 3.104 / 3.180 ms (same box, tools/ab_env.py bench:C2).  The
        default follows the mode: 70 % when this step's grouping ran ahead, 80 % in order.)"""
        if self.training and torch.is_grad_enabled() and self.chain_priority():
            ahead = getattr(self.point_encoder, "group_ahead", None) is not None
            return ops.persistent_occupancy(int(os.environ.get("PPT_TOWER_OCCUPANCY", "70" if ahead else "80")))
        if not self.training and not torch.is_grad_enabled() and (self.eval_inputs_ready or getattr(self.point_encoder, "group_ahead", None) is not None):
            # validate() with the next batch's tokenizer on its own stream: its persistent kernels leave a fifth of the CUs to
            # the blocks they run beside (C2 eval 2.44 -> 2.40 ms; 60 %: 2.43, 40 %: 2.56)
            return ops.persistent_occupancy(int(os.environ.get("PPT_EVAL_OCCUPANCY", "80")))
        if self.training and torch.is_grad_enabled() and os.environ.get("PPT_TOWER_OCCUPANCY_ALWAYS"):
            return ops.persistent_occupancy(int(os.environ["PPT_TOWER_OCCUPANCY_ALWAYS"]))       # (experiments)
        if self.training and torch.is_grad_enabled() and os.environ.get("PPT_TOWER_PRIO", "1") == "1":
            # the TOWER is the critical path (a training point side, or a long frozen encoder): its kernels that take the
            # priority argument are issued ahead of the chain's where they share a SIMD (C3 6.56 -> 6.53 ms, C5 7.69 -> 7.65 ms)
            return ops.wave_priority(1)
        return contextlib.nullcontext()

    def reset_caches(self):
        """Drop everything derived from parameter / buffer STORAGE: the state-dict views, the operand copies of the
        weights and every captured hipGraph (a graph bakes in the device pointers of what it read) -- of this module
        AND of the point encoder.  nn.Module.load_state_dict loads children through _load_from_state_dict, so the
        point encoder's own load_state_dict override never runs for a top-level load; train.BufferBroadcast re-binds
        the BatchNorm buffers to views of a flat tensor.  Both call this."""
        self._sd = None
        self._wc = None
        self._te_cache = None
        self._chain_prio = None
        self._text_calibrated = False               # new weights: the half-vs-fp32 check of the text tower runs again
        self._split_fit_pending = True              # ... and so does the range fit of the split16 pre-scale (_cache)
        if self.text_calibration is not None and self.text_calibration.get("demoted"):
            self.text_precision = None
        self.text_calibration = None
        self._graphs.clear()
        pe = getattr(self, "point_encoder", None)
        if pe is not None:
            for attr in ("_sd", "_wc"):
                if hasattr(pe, attr):
                    setattr(pe, attr, None)
            if hasattr(pe, "_graphs"):
                pe._graphs.clear()

    def _apply(self, fn, *a, **k):
        self._eot_pos = None
        self.reset_caches()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.reset_caches()
        return super().load_state_dict(*a, **k)

    def _text_len(self):
        if self._text_len_cache is None or self._text_len_cache[0] is not self.tokenized_prompts:
            self._text_len_cache = (self.tokenized_prompts, int(self.tokenized_prompts.argmax(dim=-1).max().item()) + 1)
        return self._text_len_cache[1]

    def _eot(self, device):
        if self._eot_pos is None or self._eot_pos.device != device:
            self._eot_pos = self.tokenized_prompts.argmax(dim=-1).to(device)
        return self._eot_pos

    # ---- reference API ------------------------------------------------------------------------
    def _head_precision(self):
        """fp32 head products, except the per-point head of part segmentation in the performance mode: fp16 operands (_MatmulNT;
        the logit scale is then applied to the PRODUCT, so that the operands stay far inside fp16's range)."""
        if self.task == 'partseg' and self.precision == torch.bfloat16:
            return torch.bfloat16 if "head" in self.demoted else torch.float16
        return torch.float32

    def encode_text(self, prompts, tokenized_prompts=None):
        """ULIP_models.py:203-222: prompts [C,77,W] -> [C,embed_dim].  General in `prompts`, as the reference: the shared-prefix
        evaluation is used only when the tensor's leading positions really are the same for every class."""
        return _TextTowerFn.apply(self, prompts, False)

    def _prefix_of(self, prompts, cand):
        """`cand` if positions 0 .. cand-1 of `prompts` [C,L,W] are identical in every class, else 0.  One host read per call:
        the public encode_text is not the training path (train.Trainer goes through _text_raw, whose prompts are PromptLearner's)."""
        if prompts.shape[1] < cand:
            return 0
        return cand if bool((prompts[:, :cand] == prompts[:1, :cand]).all().item()) else 0

    def encode_pc(self, pc, cls_label=None):
        """ULIP_models.py:250-258."""
        if self.task == 'partseg':
            pc_feat = self.point_encoder(pc, cls_label)
        else:
            pc_feat = self.point_encoder(pc)
        wt = engine._f32_cache(self._cache()).get(self.pc_projection, "wt")     # [embed, feat]
        lead = pc_feat.shape[:-1]
        return matmul_nt(pc_feat.reshape(-1, pc_feat.shape[-1]), wt, self._head_precision()).view(*lead, -1)

    TEXT_CALIBRATION_THRESHOLD = 1e-2

    def calibrate_text_precision(self, threshold=None, force=False):
        """The mixed 16-bit mode checks ITSELF against fp32 on the weights it was given (round 5).  Every parity
        bound of this repository was measured on std-0.02 synthetic weights, where the CLIP text tower on IEEE-half operands is within
        1e-3 of fp32.  On checkpoint-LIKE magnitudes (ppt_amd.weights.checkpoint_like: LayerNorm gains with 5-10 x outlier channels,
        3 x larger weight matrices -- attention scores of O(100)) the same tower is off by 8 % rms of the logit range and the token
        gradient by more than its own norm: finite, so health.Monitor never sees it (tools/ckpt_like_error.py: text tower 14.3 of
        |logits| <= 72; with the text tower on fp32 operands 0.49, every other stage left in half).  So, once per weight set, the text
        features of the CURRENT prompts are computed on half and on fp32 operands (two forwards of the 817-row tower, no_grad, eager)
        and compared: relative L2 distance of the L2-normalised features above `threshold` (default 1e-2; 2.6e-4 on the synthetic
        weights) -> the text tower runs on fp32 operands from here on (`text_precision = torch.float32`), with a warning.  The result
        is kept in `text_calibration`.  PPT_TEXT_CALIBRATE=0 turns the check off; set_precision / load_state_dict re-arm it."""
        if (self._text_calibrated and not force) or os.environ.get("PPT_TEXT_CALIBRATE", "1") == "0":
            return self.text_calibration
        self._text_calibrated = True
        tok = self.prompt_learner.learnable_tokens
        if self.precision != torch.bfloat16 or self.text_precision is not None or not tok.is_cuda:
            return self.text_calibration
        if torch.cuda.is_current_stream_capturing():
            self._text_calibrated = False
            return self.text_calibration
        thr = self.TEXT_CALIBRATION_THRESHOLD if threshold is None else float(threshold)
        graphs_on, self.use_hip_graphs = self.use_hip_graphs, False
        try:
            with torch.no_grad():
                lo = self._text_raw().float()
                self.text_precision, self._wc = torch.float32, None
                hi = self._text_raw().float()
        finally:
            self.use_hip_graphs = graphs_on
            self.text_precision, self._wc = None, None
        lo = lo / lo.norm(dim=-1, keepdim=True)
        hi = hi / hi.norm(dim=-1, keepdim=True)
        rel = float(((lo - hi).norm() / hi.norm()).item())
        bad = not (rel <= thr)                      # (NaN -> demote as well)
        # under a process group every rank must take the SAME decision (a rank that alone re-captures its graphs and runs a slower
        # chain stalls the others' collective): MAX over the ranks of the verdict and of the measured distance
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            v = torch.tensor([1.0 if bad else 0.0, rel if rel == rel else 1e30], dtype=torch.float64, device=tok.device)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            bad, rel = bool(v[0].item() > 0), float(v[1].item())
        self.text_calibration = {"rel_l2": rel, "threshold": thr, "demoted": bad}
        self._te_cache = None
        if bad:
            import warnings
            self.text_precision = torch.float32
            self._graphs.clear()
            if self.text_split16:
                self._fit_split16_range()
            warnings.warn(f"ppt_amd: on these weights the CLIP text tower on IEEE-half operands differs from fp32 by {rel:.3g} "
                          f"(relative L2 of the normalised text features; threshold {thr:g}): the text tower runs on fp32 operands "
                          + ("multiplied as hi + lo half pairs (split16) " if self.text_split16 else "") +
                          "from here on (slower prompt chain, reference-grade text features and token gradients).  "
                          "PPT_TEXT_CALIBRATE=0 keeps half.", RuntimeWarning, stacklevel=3)
        return self.text_calibration

    def _text_raw(self):
        """encode_text(prompt_learner()) -- through the one-kernel prompt assembly (_TextTowerTokensFn) on a GPU."""
        if not self._text_calibrated:
            self.calibrate_text_precision()
        tok = self.prompt_learner.learnable_tokens
        if self.fused_prompt_rows and tok.is_cuda and self.prompt_learner.class_name_position in ("front", "middle", "end"):
            return _TextTowerTokensFn.apply(self, tok)
        return _TextTowerFn.apply(self, self.prompt_learner(), True)        # PromptLearner's output: the prefix is shared by construction

    def _text_embed(self):
        # inference fast path: the text features depend only on the prompt tokens, so
        # validate() (main_cls.py:237-299) needs them once per epoch, not once per batch
        tok = self.prompt_learner.learnable_tokens
        if not torch.is_grad_enabled() and self._te_cache is not None and self._te_cache[0] == (tok._version, tok.data_ptr(),
                                                                                               self.precision):
            return self._te_cache[1]
        text_embed = self._text_raw()
        text_embed = text_embed / text_embed.norm(dim=-1, keepdim=True)
        if not torch.is_grad_enabled():
            self._te_cache = ((tok._version, tok.data_ptr(), self.precision), text_embed)
        return text_embed

    def text_stream(self):
        """The HIP stream the prompt side (PromptLearner + text tower, and in training its backward and the
        optimizer: train.Trainer.step) is queued on."""
        if self._text_stream is None:
            self._text_stream = graphs.shared_text_stream()
        return self._text_stream

    def forward(self, pc, cls_label=None):
        """ULIP_models.py:260-283 -> logits [B,C] (partseg: [B,N,C])."""
        # the rows the caller's criterion will average over: fixes the power-of-two scale of every 16-bit backward stage created
        # by this forward (ppt_amd/gradscale.py) -- a plain `loss.backward()` (main_cls.py:197, main_partseg.py:214) is then
        # scaled and un-scaled inside the nodes
        with gradscale.rows(pc.shape[0] * (pc.shape[1] if self.task == 'partseg' else 1)):
            return self._forward(pc, cls_label)

    def _forward(self, pc, cls_label=None):
        cur = torch.cuda.current_stream()
        side = None
        if self.overlap_text_tower and pc.is_cuda:
            side = self.text_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                text_embed = self._text_embed()
        pe = self.point_encoder
        ahead = ((self.eval_inputs_ready or graphs.ready_event(pc) is not None) and not self.training and not torch.is_grad_enabled() and pc.is_cuda
                 and hasattr(pe, "group_ahead") and self.task != 'partseg')
        if ahead:
            # validate() with resident inputs (eval_inputs_ready): the next batch's FPS + kNN + tokenizer on the grouping stream,
            # under this batch's blocks -- the same ahead stage train.Trainer uses (PointTransformer._group_ahead)
            pe.group_ahead = graphs.shared_group_stream()
            pe.inputs_vouched = bool(self.eval_inputs_ready)     # (else: only tensors that carry their copy event run ahead)
        try:
            with self._tower_room():
                pc_embed = self.encode_pc(pc, cls_label) if self.task == 'partseg' else self.encode_pc(pc)
        finally:
            if ahead:
                pe.group_ahead = None
                pe.inputs_vouched = False
        if self.health is not None and self.training:
            self.health.check(1, pc_embed)                          # (BIT_POINT)
        if side is not None:
            cur.wait_stream(side)
            text_embed.record_stream(cur)
        else:
            text_embed = self._text_embed()
        logit_scale = self.logit_scale.exp()
        lead = pc_embed.shape[:-1]
        hp = self._head_precision()
        if hp in ops.HALF:            # ULIP_models.py:281 with the scale moved behind the product (same value, fp16-safe operands)
            logits = logit_scale * matmul_nt(pc_embed.reshape(-1, pc_embed.shape[-1]), text_embed, hp)
        else:
            logits = matmul_nt((logit_scale * pc_embed).reshape(-1, pc_embed.shape[-1]), text_embed, hp)
        return logits.view(*lead, -1)

    def forward_loss(self, pc, labels, smoothing):
        """forward + CrossEntropyLoss(label_smoothing) for the case where nothing on the point side trains (head_type 0):
        same two-stream schedule as forward(), but everything between the towers is one fused, graph-replayed node
        (_HeadLossFn) instead of ~25 autograd-tracked launches.  -> (loss, logits)."""
        with gradscale.rows(pc.shape[0]):
            return self._forward_loss(pc, labels, smoothing)

    def _forward_loss(self, pc, labels, smoothing):
        cur = torch.cuda.current_stream()
        side = self.text_stream() if (self.overlap_text_tower and pc.is_cuda) else None
        if side is not None:
            side.wait_stream(cur)
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            self._handoff = self._handoff_grad and self.use_hip_graphs
            try:
                text_raw = self._text_raw()
            finally:
                self._handoff = False
        tw = getattr(self, "tower_stream", None) if side is not None else None
        if tw is not None:
            # the caller vouched that `pc` is complete in memory (train.Trainer.inputs_ready): the frozen point tower runs on
            # a stream of its own that waits for nothing -- the caller's stream, which stalls at the head until the prompt
            # chain of the previous iteration has finished, then no longer holds the NEXT iteration's tower back
            with torch.cuda.stream(tw), self._tower_room():
                pc_feat = self.point_encoder(pc)
            cur.wait_stream(tw)
            pc_feat.record_stream(cur)
        else:
            with self._tower_room():
                pc_feat = self.point_encoder(pc)
        if self.health is not None:
            self.health.check(1, pc_feat)                           # (BIT_POINT: on the caller's stream, beside the prompt chain)
        if side is not None and HEAD_ON_TEXT_STREAM:
            # The head (projection, logits, loss, d loss / d text features: ~75 us of kernels) on the TEXT stream, between the text
            # forward and the text backward it sits between anyway: the prompt chain -- the step's critical path -- then never
            # leaves its stream.  On the caller's stream the chain crossed streams twice per step (text forward -> event -> head ->
            # event -> text backward: ~195 us of wall for those 75 us in tools/chain_sequence.py).  The caller's stream waits for the
            # head's results exactly where it used to produce them.
            side.wait_stream(cur)                                   # the tower's features
            with torch.cuda.stream(side):
                loss, logits = _HeadLossFn.apply(self, pc_feat, text_raw, labels, smoothing)
            pc_feat.record_stream(side)
            labels.record_stream(side)
            cur.wait_stream(side)
            loss.record_stream(cur)
            logits.record_stream(cur)
            return loss, logits
        if side is not None:
            cur.wait_stream(side)
            text_raw.record_stream(cur)
        return _HeadLossFn.apply(self, pc_feat, text_raw, labels, smoothing)


def get_metric_names():
    """ULIP_models.py:290-291."""
    return ['loss', 'acc']


def cfg_from_yaml_file(cfg_file):
    """data/dataset_3d.py:841-847 equivalent for the model yaml: nested attribute namespace."""
    import yaml

    def wrap(d):
        return SimpleNamespace(**{k: wrap(v) if isinstance(v, dict) else v for k, v in d.items()})
    with open(cfg_file) as f:
        return wrap(yaml.load(f, Loader=yaml.FullLoader))


POINTBERT_CONFIG = SimpleNamespace(NAME="PointTransformer", trans_dim=384, depth=12, drop_path_rate=0.1, cls_dim=40,
                                   num_heads=6, group_size=32, num_group=512, encoder_dims=256)
"""models/pointbert/PointTransformer_8192point.yaml:15-25 (`model:` section)."""


def unfreeze_list(head_type):
    """ULIP_models.py:461-470."""
    p = 'point_encoder.blocks.blocks.11.'
    mods = []
    if head_type > 0:
        mods += [p + 'norm2.weight', p + 'norm2.bias', p + 'mlp.fc2.weight', p + 'mlp.fc2.bias']
    if head_type > 1:
        mods += [p + 'norm1.weight', p + 'norm1.bias', p + 'mlp.fc1.weight', p + 'mlp.fc1.bias']
    if head_type > 2:
        mods += [p + 'attn.qkv.weight', p + 'attn.proj.weight', p + 'attn.proj.bias']
    return mods


SLIP_CKPT = './data/initialize_models/slip_base_100ep.pt'


def synthetic_weights_allowed(args):
    """No ULIP / SLIP checkpoint exists offline (benches, tests): `args.synthetic_weights = True` or
    PPT_SYNTHETIC_WEIGHTS=1 is the explicit opt-in to build a model WITHOUT loading them (the caller then loads a
    state dict of its own, ppt_amd/weights.py).  Without it a missing file raises, as the reference's torch.load does."""
    return bool(getattr(args, "synthetic_weights", False)) or os.environ.get("PPT_SYNTHETIC_WEIGHTS") == "1"


def read_pretrained(path):
    """`torch.load(path)['state_dict']` with the DistributedDataParallel prefix removed (ULIP_models.py:477-485)."""
    sd = torch.load(path, map_location=torch.device('cpu'))['state_dict']
    return {k.replace('module.', ''): v for k, v in sd.items()}


def _load_and_freeze(model, args, point_ckpt, skip):
    """ULIP_models.py:472-507: every parameter except the prompt tokens, a `point_encoder.cls_head` and `skip` is
    frozen and overwritten with the pretrained value -- looked up in the point checkpoint first, else in the SLIP
    checkpoint (KeyError when neither has it, as in the reference).  Un-frozen parameters are NOT loaded (SURVEY.md
    App. A Q4).  A missing checkpoint file raises FileNotFoundError like the reference's torch.load, unless the caller
    opted into synthetic weights (synthetic_weights_allowed); the two files are then looked at independently."""
    synthetic = synthetic_weights_allowed(args)
    tables = []
    for path in (point_ckpt, SLIP_CKPT):
        if os.path.exists(path) or not synthetic:
            tables.append(read_pretrained(path))            # FileNotFoundError when absent
    for name, param in model.named_parameters():
        if name == 'prompt_learner.learnable_tokens' or 'point_encoder.cls_head' in name or name in skip:
            continue
        param.requires_grad = False
        src = next((t[name] for t in tables if name in t), None)
        if src is None:
            if not synthetic:
                raise KeyError(name)
            continue
        param.data.copy_(src.data if isinstance(src, nn.Parameter) else src)
    if hasattr(model, "reset_caches"):
        model.reset_caches()


def ULIP_PointBERT_partseg(args):
    """ULIP_models.py:515-569: PointBERT part-segmentation encoder/decoder (pc_feat_dims 128).  Trainable = the
    prompt tokens + every point-encoder parameter that the ULIP PointBERT checkpoint does not contain (the
    decoder); nothing is copied from the checkpoint here, exactly as in the reference."""
    from .pointbert.point_encoder import PointTransformer, PointTransformer_partseg
    here = os.path.dirname(os.path.abspath(__file__))
    config_addr = os.path.join(os.path.dirname(here), 'models/pointbert/PointTransformer_8192point.yaml')
    config = cfg_from_yaml_file(config_addr).model if os.path.exists(config_addr) else POINTBERT_CONFIG
    point_encoder = PointTransformer_partseg(config, args=args)
    model = ULIP_WITH_IMAGE(embed_dim=512, point_encoder=point_encoder, context_length=77, vocab_size=49408,
                            classnames=args.classnames, template_init=args.template_init,
                            class_name_position=args.class_name_position,
                            num_learnable_prompt_tokens=args.num_learnable_prompt_tokens, transformer_width=512,
                            transformer_heads=8, transformer_layers=12, pc_feat_dims=128, device=args.gpu, task=args.task)
    if not getattr(args, "evaluate_3d", False):
        proj = os.path.dirname(here)
        ckpt = os.path.join(proj, 'data/pretrained_models/pointbert_ulip2.pt' if getattr(args, "ulip2", False)
                            else 'data/pretrained_models/pointbert.pt')
        if os.path.exists(ckpt) or not synthetic_weights_allowed(args):
            # the reference reads both files (FileNotFoundError when absent) and uses the point checkpoint for key
            # membership only: no value is copied (ULIP_models.py:538-565)
            backbone = set(read_pretrained(ckpt))
            read_pretrained(os.path.join(proj, SLIP_CKPT[2:]))
        else:
            backbone = {"point_encoder." + k for k in PointTransformer(config, args=args).state_dict()}   # keys of the ULIP ckpt
        for name, param in model.named_parameters():
            if name.startswith('prompt_learner'):
                continue
            if name.startswith('point_encoder.') and name not in backbone:
                continue
            param.requires_grad = False
    return model


def ULIP_PN_MSG(args):
    """ULIP_models.py:347-391: PointNet2-MSG point encoder (pc_feat_dims 256); everything except
    `prompt_learner.learnable_tokens` is frozen."""
    from .pointnet2.pointnet2 import Pointnet2_Msg
    point_encoder = Pointnet2_Msg()
    model = ULIP_WITH_IMAGE(embed_dim=512, point_encoder=point_encoder, context_length=77, vocab_size=49408,
                            classnames=args.classnames, template_init=args.template_init,
                            class_name_position=args.class_name_position,
                            num_learnable_prompt_tokens=args.num_learnable_prompt_tokens, transformer_width=512,
                            transformer_heads=8, transformer_layers=12, pc_feat_dims=256, device=args.gpu, task=args.task)
    if not getattr(args, "evaluate_3d", False):
        _load_and_freeze(model, args, './data/pretrained_models/pointnet2_msg_1kpts.pt', set())
    return model


def ULIP_PN_SSG(args):
    """ULIP_models.py:294-342: PointNet2-SSG point encoder (pc_feat_dims 256); everything except
    `prompt_learner.learnable_tokens` is frozen."""
    from .pointnet2.pointnet2 import Pointnet2_Ssg
    point_encoder = Pointnet2_Ssg()
    model = ULIP_WITH_IMAGE(embed_dim=512, point_encoder=point_encoder, context_length=77, vocab_size=49408,
                            classnames=args.classnames, template_init=args.template_init,
                            class_name_position=args.class_name_position,
                            num_learnable_prompt_tokens=args.num_learnable_prompt_tokens, transformer_width=512,
                            transformer_heads=8, transformer_layers=12, pc_feat_dims=256, device=args.gpu, task=args.task)
    if not getattr(args, "evaluate_3d", False):
        _load_and_freeze(model, args, './data/pretrained_models/pointnet2_ssg.pt', set())
    return model


def ULIP_PN_MLP(args):
    """ULIP_models.py:394-441: PointMLP point encoder (pc_feat_dims 256); everything except
    `prompt_learner.learnable_tokens` is frozen."""
    from .pointmlp.pointMLP import pointMLP
    point_encoder = pointMLP()
    model = ULIP_WITH_IMAGE(embed_dim=512, point_encoder=point_encoder, context_length=77, vocab_size=49408,
                            classnames=args.classnames, template_init=args.template_init,
                            class_name_position=args.class_name_position,
                            num_learnable_prompt_tokens=args.num_learnable_prompt_tokens, transformer_width=512,
                            transformer_heads=8, transformer_layers=12, pc_feat_dims=256, device=args.gpu, task=args.task)
    if not getattr(args, "evaluate_3d", False):
        _load_and_freeze(model, args, './data/pretrained_models/pointmlp.pt', set())
    return model


def ULIP_PointBERT(args):
    """ULIP_models.py:443-512.  args: classnames, template_init, class_name_position,
    num_learnable_prompt_tokens, gpu, task, head_type, evaluate_3d, ulip2."""
    from .pointbert.point_encoder import PointTransformer
    config_addr = './models/pointbert/PointTransformer_8192point.yaml'
    config = cfg_from_yaml_file(config_addr).model if os.path.exists(config_addr) else POINTBERT_CONFIG
    point_encoder = PointTransformer(config, args=args)
    model = ULIP_WITH_IMAGE(embed_dim=512, point_encoder=point_encoder, context_length=77, vocab_size=49408,
                            classnames=args.classnames, template_init=args.template_init,
                            class_name_position=args.class_name_position,
                            num_learnable_prompt_tokens=args.num_learnable_prompt_tokens, transformer_width=512,
                            transformer_heads=8, transformer_layers=12, pc_feat_dims=768, device=args.gpu,
                            task=args.task)
    if not getattr(args, "evaluate_3d", False):
        ckpt = './data/pretrained_models/pointbert_ulip2.pt' if getattr(args, "ulip2", False) \
            else './data/pretrained_models/pointbert.pt'
        _load_and_freeze(model, args, ckpt, set(unfreeze_list(args.head_type)))
    params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print('\n====================\n\tNumber of learnable params:', params, '\n====================\n')
    return model
