"""Import harness for the upstream reference (auniquesun/PPT) -- BUILD CONTAINER ONLY.

The reference lives read-only at /root/reference and never travels to the GPU box; this
module is used solely by tests/golden/make_golden.py (fixture generation) and by the
optional `-m "not gpu"` cross-checks that skip when /root/reference is absent.

It follows SURVEY.md App. C: third-party modules that are not installed here are replaced
by minimal stand-ins *in sys.modules only* (nothing is written into the reference tree),
and `Tensor.cuda` is made the identity because PromptLearner.__init__ calls `.cuda()`
unconditionally (reference models/ULIP_models.py:102).
"""
import os
import sys
import types
import contextlib

REF_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "models", "pointbert"))


class _AttrDict(dict):
    """easydict.EasyDict stand-in: recursive attribute dictionary."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            v = _AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _install_stubs():
    import torch
    import torch.nn as nn

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):
        """timm==0.4.12 DropPath: x.div(keep) * floor(keep + U[0,1)) per sample, train only."""

        def __init__(self, drop_prob=None):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if not self.drop_prob or not self.training:
                return x
            keep = 1.0 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            mask = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
            mask.floor_()
            return x.div(keep) * mask

    if "timm" not in sys.modules:
        mod("timm")
        mod("timm.models")
        mod("timm.models.layers", DropPath=DropPath)
    if "termcolor" not in sys.modules:
        mod("termcolor", colored=lambda s, *a, **k: s)
    if "easydict" not in sys.modules:
        mod("easydict", EasyDict=_AttrDict)
    for name in ("h5py", "open3d"):
        if name not in sys.modules:
            mod(name)
    if "ftfy" not in sys.modules:
        mod("ftfy", fix_text=lambda s: s)
    if "cosine_annealing_warmup" not in sys.modules:
        mod("cosine_annealing_warmup", CosineAnnealingWarmupRestarts=object)
    if "torch._six" not in sys.modules:
        mod("torch._six", string_classes=(str,))
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self


@contextlib.contextmanager
def reference_context():
    """chdir into the reference (yaml / labels.json paths are cwd-relative) with it on sys.path."""
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    _install_stubs()
    old = os.getcwd()
    # our repo also has a top-level `models` shim; make sure the reference's wins in here
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")
             or k == "utils" or k.startswith("utils.") or k == "data" or k.startswith("data.")}
    for k in saved:
        del sys.modules[k]
    # the reference's `models` is a namespace package (no __init__.py); this repo's `models/` alias package is a
    # regular one and would win from ANY position on sys.path -- hide the repo root while the reference imports
    repo_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    hidden = [q for q in sys.path if os.path.abspath(q or old) == repo_root]
    saved_path = list(sys.path)
    sys.path[:] = [REF_ROOT] + [q for q in sys.path if q not in hidden]
    os.chdir(REF_ROOT)
    try:
        yield
    finally:
        os.chdir(old)
        sys.path[:] = saved_path
        ref_mods = [k for k in sys.modules if k == "models" or k.startswith("models.")
                    or k == "utils" or k.startswith("utils.") or k == "data" or k.startswith("data.")]
        for k in ref_mods:
            sys.modules["_ref_" + k] = sys.modules.pop(k)
        sys.modules.update(saved)


def build_reference_ulip_pointbert(classnames, head_type=0, class_name_position="middle",
                                   num_learnable_prompt_tokens=32, task="cls"):
    """Construct the reference ULIP_WITH_IMAGE + PointTransformer exactly as
    models/ULIP_models.py:443-459 does, then apply the freeze list of :461-507 by name
    (pretrained checkpoints are not available, so nothing is loaded)."""
    import argparse
    with reference_context():
        import models.ULIP_models as M
        from models.pointbert.point_encoder import PointTransformer
        cfg = M.cfg_from_yaml_file("./models/pointbert/PointTransformer_8192point.yaml")
        ns = argparse.Namespace(head_type=head_type)
        pe = PointTransformer(cfg.model, args=ns)
        m = M.ULIP_WITH_IMAGE(embed_dim=512, point_encoder=pe, context_length=77, vocab_size=49408,
                              classnames=classnames, template_init="",
                              class_name_position=class_name_position,
                              num_learnable_prompt_tokens=num_learnable_prompt_tokens,
                              transformer_width=512, transformer_heads=8, transformer_layers=12,
                              pc_feat_dims=768, device=0, task=task)
    unfreeze = []
    p = "point_encoder.blocks.blocks.11."
    if head_type > 0:
        unfreeze += [p + "norm2.weight", p + "norm2.bias", p + "mlp.fc2.weight", p + "mlp.fc2.bias"]
    if head_type > 1:
        unfreeze += [p + "norm1.weight", p + "norm1.bias", p + "mlp.fc1.weight", p + "mlp.fc1.bias"]
    if head_type > 2:
        unfreeze += [p + "attn.qkv.weight", p + "attn.proj.weight", p + "attn.proj.bias"]
    for name, param in m.named_parameters():
        if name == "prompt_learner.learnable_tokens" or name in unfreeze:
            continue
        param.requires_grad = False
    return m
