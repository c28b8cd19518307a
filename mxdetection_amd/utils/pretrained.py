"""ImageNet ResNet weights from an MXNet `.params` file into the backbone (SURVEY.md section 8f rank 1;
/root/reference/README.md:25,37). Frozen BatchNorm (use_global_stats) is folded into the preceding filter and a
per-channel shift (utils.params_io.fold_batchnorm), which is the form every conv kernel here consumes.

Symbol naming assumed ("mx_resnet_v1": the resnet-v1-50/101 symbols the MXNet detection lineage pretrains from --
the reference ships no symbol file, so the table below is this repo's reading of that convention; pass your own
`names` iterable for another one):
    conv0_weight, bn0_{gamma,beta,moving_mean,moving_var}
    stage{S}_unit{U}_conv{1,2,3}_weight, stage{S}_unit{U}_bn{1,2,3}_*
    stage{S}_unit1_sc_weight, stage{S}_unit1_sc_bn_*            (projection shortcut)
with "arg:" / "aux:" key prefixes as written by mx.model.save_checkpoint (bare names are accepted too).
"""
import numpy as np

from .params_io import fold_batchnorm, load_params

_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


def resnet_v1_names(depth):
    """(our layer name, MXNet conv weight name, MXNet BatchNorm prefix) for every backbone convolution."""
    yield "stem", "conv0_weight", "bn0"
    for si, nb in enumerate(_BLOCKS[depth]):
        for bi in range(nb):
            u, ours = "stage%d_unit%d" % (si + 1, bi + 1), "layer%d.%d" % (si + 1, bi)
            for k in (1, 2, 3):
                yield "%s.conv%d" % (ours, k), "%s_conv%d_weight" % (u, k), "%s_bn%d" % (u, k)
            if bi == 0:
                yield ours + ".down", u + "_sc_weight", u + "_sc_bn"


def _get(blob, name):
    for k in ("arg:" + name, "aux:" + name, name):
        if k in blob:
            return np.asarray(blob[k])
    return None


def load_pretrained_backbone(model, params, depth=50, names=None, eps=2e-5, fix_gamma=False, strict=True):
    """params: path to a .params file or an already loaded {name: ndarray}. Returns the list of missing MXNet names."""
    import torch
    blob = load_params(params) if isinstance(params, (str, bytes)) else params
    if not isinstance(blob, dict):
        raise ValueError("the .params file carries no names (NDArray list), cannot map it onto the backbone")
    tensors = {n: t for n, _, t, _ in model._named_tensors()}
    missing = []
    for ours, wname, bn in (names or resnet_v1_names(depth)):
        w = _get(blob, wname)
        stats = [_get(blob, bn + s) for s in ("_gamma", "_beta", "_moving_mean", "_moving_var")]
        if w is None or any(s is None for s in stats[1:]) or (stats[0] is None and not fix_gamma):
            missing.append(wname if w is None else bn)
            continue
        wf, bf = fold_batchnorm(w, stats[0], stats[1], stats[2], stats[3], eps=eps, fix_gamma=fix_gamma)
        tw, tb = tensors[ours + ".weight"], tensors[ours + ".bias"]
        src = torch.from_numpy(wf).permute(0, 2, 3, 1).contiguous()             # OIHW -> [Cout,KH,KW,Cin]
        if tuple(src.shape) != tuple(tw.shape):
            raise ValueError("%s: %s has shape %s, the model expects %s" % (ours, wname, tuple(src.shape), tuple(tw.shape)))
        tw.copy_(src.to(tw.device).to(tw.dtype))
        tb.copy_(torch.from_numpy(bf).to(tb.device).to(tb.dtype))
    if strict and missing:
        raise KeyError("pretrained file lacks %d backbone tensors, e.g. %s" % (len(missing), missing[:3]))
    model.arena.refresh_bf16()
    model.refresh_transposed()
    return missing
