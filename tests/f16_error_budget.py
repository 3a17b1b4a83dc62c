"""Dev script (not a test): CPU emulation of the half-precision plan's roundings on top of the oracle's forward, to
find which stored tensors carry the EPE of the fp16 path.  Every conv accumulates in fp64 here, so
the only error sources are the roundings switched on below.

    python tests/f16_error_budget.py [s|m|full|kitti]
    python tests/f16_error_budget.py [s|m|m2|m3|full|kitti] policies [codes...]      (round 3: per-block precision policies)

Switches (all True = the round-1 plan):
  w      filters rounded to half                         act    conv outputs (trunk activations) rounded to half
  corr   cost volume rounded to half                     warp   warped features rounded to half
  head   predict_flow outputs rounded to half            upflow deconv output (up_flow) rounded to half before the warp
  flowin up_flow/up_feat rounded to half as conv inputs  headw  head / deconv filters rounded to half
"""
import os
import re
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pwc_oracle as O                                   # noqa: E402
from opticalflow_amd.weights import synthetic_state_dict             # noqa: E402


def q(x):
    return x.half().to(x.dtype)


def forward(sd, x, cfg, all_levels=False):
    dt = torch.float64
    x = x.to(dt)

    def qif(flag, t):
        return q(t) if cfg[flag] else t

    groups = cfg.get("groups")

    def group_of(name):
        if name.startswith("dc_"):
            return "ctx"
        m = re.match(r"conv(\d)_", name)
        return "dec" + m.group(1) if m else "pyr"

    def conv(name, t, stride=1, dilation=1, act=True, wflag="w", oflag="act", first=False):
        key = name + ".0" if (name + ".0.weight") in sd else name
        w = sd[key + ".weight"].to(dt)
        if groups is not None and wflag == "w" and group_of(name) not in groups:
            wflag = oflag = "nop"
        if not first:
            w = qif(wflag, w)
        y = F.conv2d(t, w, sd[key + ".bias"].to(dt), stride=stride, padding=dilation, dilation=dilation)
        if act:
            y = O.leaky_relu(y)
        return qif(oflag, y)

    def deconv(name, t, wflag, oflag):
        w = qif(wflag, sd[name + ".weight"].to(dt))
        return qif(oflag, F.conv_transpose2d(t, w, sd[name + ".bias"].to(dt), stride=2, padding=1))

    feats = []
    for im in (x[:, :3], x[:, 3:]):
        pyr, t = [], im
        for i, (name, stride) in enumerate(O.PYRAMID):
            t = conv(name, t, stride=stride, first=(i == 0))
            if i % 3 == 2:
                pyr.append(t)
        feats.append(pyr)
    flows = {}
    up_flow = up_feat = None
    for lvl in (6, 5, 4, 3, 2):
        c1, c2 = feats[0][lvl - 1], feats[1][lvl - 1]
        if lvl == 6:
            xcat = qif("corr", O.leaky_relu(O.correlation(c1, c2, 4, 1, 4, 1, 1, 1)))
        else:
            wf = qif("upflow", up_flow)
            w = qif("warp", O.warp(c2, wf * O.WARP_SCALE[lvl]))
            corr = qif("corr", O.leaky_relu(O.correlation(c1, w, 4, 1, 4, 1, 1, 1)))
            xcat = torch.cat((corr, c1, qif("flowin", up_flow), qif("flowin", up_feat)), 1)
        for i in range(5):
            xcat = torch.cat((conv("conv%d_%d" % (lvl, i), xcat), xcat), 1)
        flow = conv("predict_flow%d" % lvl, xcat, act=False, wflag="headw", oflag="head")
        flows[lvl] = flow
        if lvl > 2:
            up_flow = deconv("deconv%d" % lvl, flow, "headw", "nop")
            up_feat = deconv("upfeat%d" % lvl, xcat, "headw", "nop")
    t = xcat
    for i, dil in enumerate(O.DILATIONS):
        t = conv("dc_conv%d" % (i + 1), t, dilation=dil)
    flow2 = flows[2] + conv("dc_conv7", t, act=False, wflag="headw", oflag="head")
    if all_levels:
        return flow2, flows[3], flows[4], flows[5], flows[6]
    return flow2


# ---- round 3: per-block precision POLICIES (the strict mode's design study) ---------------------------------------------------------
# A policy gives every block -- "pyr", "dec6" .. "dec2", "ctx" -- one of
#   h  half filters and half activations (the fast fp16 plan)          s  split (hi + lo) filters, half activations
#   f  fp32 block: nothing rounded inside; what it hands to a half block is rounded at the hand-over
# The flow chain is fp32 in every policy (heads / deconvs unrounded), as in the shipped plan.  Hand-overs: the pyramid features c1_l,
# c2_l are half wherever decoder level l is a half block (correlation / warp inputs and the c1 slot of the arena); the cost volume,
# up_flow / up_feat copies are half wherever they are conv inputs of a half block.
def forward_policy(sd, x, pol, all_levels=False):
    dt = torch.float64
    x = x.to(dt)

    def mode_of(name):
        if name.startswith("dc_"):
            return pol["ctx"]
        m = re.match(r"conv(\d)_", name)
        return pol["dec" + m.group(1)] if m else pol["pyr"]

    def conv(name, t, stride=1, dilation=1, act=True, first=False, mode=None):
        key = name + ".0" if (name + ".0.weight") in sd else name
        mode = mode or pol.get("layer_mode", {}).get(name) or mode_of(name)      # layer_mode: per-layer override of the block's policy
        w = sd[key + ".weight"].to(dt)
        if mode == "h" and not first:
            w = q(w)
        y = F.conv2d(t, w, sd[key + ".bias"].to(dt), stride=stride, padding=dilation, dilation=dilation)
        if act:
            y = O.leaky_relu(y)
        # round-3 what-if switches: "noact": layer-name prefixes whose stored outputs keep full precision (hi + lo storage)
        if any(name.startswith(pre) for pre in pol.get("noact", ())):
            return y
        return q(y) if mode in ("h", "s") else y

    feats = []
    for im in (x[:, :3], x[:, 3:]):
        pyr, t = [], im
        for i, (name, stride) in enumerate(O.PYRAMID):
            t = conv(name, t, stride=stride, first=(i == 0))
            if i % 3 == 2:
                pyr.append(t)
        feats.append(pyr)
    flows = {}
    up_flow = up_feat = None
    for lvl in (6, 5, 4, 3, 2):
        half = pol["dec%d" % lvl] in ("h", "s")
        hq = q if half else (lambda t: t)
        if lvl == 2 and pol.get("base2_exact"):          # what-if: the level-2 base channels (corr, c1, up_flow, up_feat) stored hi + lo
            hq = lambda t: t
        # what-if: only SOME of the level-2 base channels stored hi + lo ("corr", "c1", "flow")
        parts = pol.get("base2_exact_parts", ()) if lvl == 2 else ()
        hq_c1 = (lambda t: t) if "c1" in parts else hq
        hq_corr = (lambda t: t) if "corr" in parts else hq
        hq_flow = (lambda t: t) if "flow" in parts else hq
        c1f, c2 = feats[0][lvl - 1], hq(feats[1][lvl - 1])
        c1 = hq_c1(c1f)
        if lvl == 6:
            xcat = hq(O.leaky_relu(O.correlation(c1, c2, 4, 1, 4, 1, 1, 1)))
        else:
            w = O.warp(c2, up_flow * O.WARP_SCALE[lvl])             # fused warp+correlation: the warped features are never stored
            corr = hq_corr(O.leaky_relu(O.correlation(hq(c1f) if lvl != 2 or not parts else c1f, w, 4, 1, 4, 1, 1, 1)))
            xcat = torch.cat((corr, c1, hq_flow(up_flow), hq_flow(up_feat)), 1)
        for i in range(5):
            xcat = torch.cat((conv("conv%d_%d" % (lvl, i), xcat), xcat), 1)
        flow = conv("predict_flow%d" % lvl, xcat, act=False, mode="f")
        flows[lvl] = flow
        if lvl > 2:
            up_flow = F.conv_transpose2d(flow, sd["deconv%d.weight" % lvl].to(dt), sd["deconv%d.bias" % lvl].to(dt), stride=2, padding=1)
            up_feat = F.conv_transpose2d(xcat, sd["upfeat%d.weight" % lvl].to(dt), sd["upfeat%d.bias" % lvl].to(dt), stride=2, padding=1)
    t = xcat if pol["ctx"] == "f" or pol["dec2"] != "f" else xcat       # (a half ctx block reading an fp32 dec2 arena would round it)
    if pol["ctx"] != "f" and pol["dec2"] == "f":
        t = q(xcat)
    for i, dil in enumerate(O.DILATIONS):
        t = conv("dc_conv%d" % (i + 1), t, dilation=dil)
    flow2 = flows[2] + conv("dc_conv7", t, act=False, mode="f")
    if all_levels:
        return flow2, flows[3], flows[4], flows[5], flows[6]
    return flow2


POLICIES = [   # name, pyr, dec6, dec5, dec4, dec3, dec2, ctx
    ("fast plan: everything half", "hhhhhhh"),
    ("split filters everywhere", "sssssss"),
    ("pyr f32, rest half", "fhhhhhh"),
    ("pyr+dec6..3 f32, dec2+ctx half", "ffffhhh"),
    ("pyr+dec6..3 f32, dec2+ctx split", "fffffss"[:4] + "fss"),
    ("pyr f32, dec6..4 half, dec3 f32, dec2+ctx split", "fhhhfss"),
    ("pyr f32, dec6..3 split, dec2+ctx split", "fssssss"),
    ("pyr split, all split", "sssssss"),
    ("pyr f32, dec6..4 half, dec3+dec2+ctx split", "fhhhsss"),
    ("pyr+dec6..3 f32, dec2 split, ctx half", "fffffsh"),
    ("pyr+dec6..3 f32, dec2 half, ctx split", "fffffhs"),
    ("all f32 but ctx split", "ffffffs"),
    ("all f32 but dec2 split", "fffffsf"),
]


WHATIF = [   # on top of fffffss
    ("fffffss", {}),
    ("+ level-2 base channels exact", {"base2_exact": True}),
    ("+ conv2_* outputs exact", {"noact": ("conv2_",)}),
    ("+ dc_conv* outputs exact", {"noact": ("dc_conv",)}),
    ("+ base exact, conv2_0/1 exact", {"base2_exact": True, "noact": ("conv2_0", "conv2_1")}),
    ("+ base exact + dc_conv1 exact", {"base2_exact": True, "noact": ("dc_conv1",)}),
    ("+ base + conv2_* exact", {"base2_exact": True, "noact": ("conv2_",)}),
    ("+ only corr of the base exact", {"base2_exact_parts": ("corr",)}),
    ("+ only c1 of the base exact", {"base2_exact_parts": ("c1",)}),
    ("+ only up_flow / up_feat exact", {"base2_exact_parts": ("flow",)}),
    ("+ corr and c1 exact", {"base2_exact_parts": ("corr", "c1")}),
    # the shipped strict plan (corr + flow residuals) and cheaper variants: which layers need their split (hi + lo) filters?
    ("shipped: corr + flow residuals", {"base2_exact_parts": ("corr", "flow")}),
    ("shipped, dc_conv2..6 plain half filters", {"base2_exact_parts": ("corr", "flow"), "layer_mode": {"dc_conv%d" % i: "h" for i in range(2, 7)}}),
    ("shipped, dc_conv4..6 plain half filters", {"base2_exact_parts": ("corr", "flow"), "layer_mode": {"dc_conv%d" % i: "h" for i in range(4, 7)}}),
    ("shipped, conv2_3 conv2_4 plain half filters", {"base2_exact_parts": ("corr", "flow"), "layer_mode": {"conv2_3": "h", "conv2_4": "h"}}),
    ("shipped, only dc_conv1 conv2_0 conv2_1 split", {"base2_exact_parts": ("corr", "flow"), "layer_mode": dict({"dc_conv%d" % i: "h" for i in range(2, 7)}, conv2_2="h", conv2_3="h", conv2_4="h")}),
    ("shipped, all level-2 / context filters plain half", {"base2_exact_parts": ("corr", "flow"), "layer_mode": dict({"dc_conv%d" % i: "h" for i in range(1, 7)}, conv2_0="h", conv2_1="h", conv2_2="h", conv2_3="h", conv2_4="h")}),
]


def main_whatif(which):
    shape, seed = {"s": ((1, 6, 64, 64), 1234), "m": ((2, 6, 128, 192), 1235), "full": ((1, 6, 448, 1024), 1234),
                   "kitti": ((1, 6, 384, 1280), 77)}[which]
    x = torch.rand(shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float32)
    sd = synthetic_state_dict(O.state_dict_manifest(), seed=0, gain=0.85, bias_std=0.02)
    torch.set_num_threads(8)
    names = ("pyr", "dec6", "dec5", "dec4", "dec3", "dec2", "ctx")
    with torch.no_grad():
        ref = forward_policy(sd, x, dict(zip(names, "fffffff")))
        print("== %s  mean|flow2| %.3f" % (which, ref.abs().mean().item()), flush=True)
        only = sys.argv[3:]
        for name, extra in WHATIF:
            if only and not any(o in name for o in only):
                continue
            pol = dict(zip(names, "fffffss"))
            pol.update(extra)
            print("%-40s EPE flow2 %.3e" % (name, O.epe(forward_policy(sd, x, pol), ref)), flush=True)


def main_policies(which):
    shape, seed = {"s": ((1, 6, 64, 64), 1234), "m": ((2, 6, 128, 192), 1235), "full": ((1, 6, 448, 1024), 1234),
                   "kitti": ((1, 6, 384, 1280), 77), "m2": ((2, 6, 128, 192), 4321), "m3": ((2, 6, 192, 256), 99)}[which]
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(shape, generator=g, dtype=torch.float32)
    sd = synthetic_state_dict(O.state_dict_manifest(), seed=0, gain=0.85, bias_std=0.02)
    torch.set_num_threads(8)
    names = ("pyr", "dec6", "dec5", "dec4", "dec3", "dec2", "ctx")
    only = sys.argv[3:]
    with torch.no_grad():
        ref = forward_policy(sd, x, dict(zip(names, "fffffff")), all_levels=True)
        print("== %s  mean|flow2| %.3f" % (which, ref[0].abs().mean().item()), flush=True)
        for name, code in POLICIES:
            if only and code not in only:
                continue
            out = forward_policy(sd, x, dict(zip(names, code)), all_levels=True)
            print("%-52s %s EPE flow2 %.3e | " % (name, code, O.epe(out[0], ref[0])) +
                  " ".join("L%d %.2e" % (l, O.epe(o, r)) for l, o, r in zip((3, 4, 5, 6), out[1:], ref[1:])), flush=True)


ALL = ("w", "act", "corr", "warp", "head", "upflow", "flowin", "headw")


def cfg_of(on):
    c = {k: (k in on) for k in ALL}
    c["nop"] = False
    return c


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "s"
    if len(sys.argv) > 2 and sys.argv[2] == "policies":
        return main_policies(which)
    if len(sys.argv) > 2 and sys.argv[2] == "whatif":
        return main_whatif(which)
    shape, seed = {"s": ((1, 6, 64, 64), 1234), "m": ((2, 6, 128, 192), 1235), "full": ((1, 6, 448, 1024), 1234),
                   "kitti": ((1, 6, 384, 1280), 77)}[which]
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(shape, generator=g, dtype=torch.float32)
    sd = synthetic_state_dict(O.state_dict_manifest(), seed=0, gain=0.85, bias_std=0.02)
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = forward(sd, x, cfg_of(()), all_levels=True)
        print("mean|flow2| %.3f" % ref[0].abs().mean().item())
        runs = [("round-1 plan (everything half)", ALL),
                ("only w", ("w",)), ("only act", ("act",)), ("only corr", ("corr",)), ("only warp", ("warp",)),
                ("only head", ("head",)), ("only upflow", ("upflow",)), ("only flowin", ("flowin",)), ("only headw", ("headw",)),
                ("fp32 flow chain (head, upflow, headw off)", ("w", "act", "corr", "warp", "flowin")),
                ("... + warp off (fused into corr)", ("w", "act", "corr", "flowin")),
                ("... + corr off", ("w", "act", "flowin")),
                ("w + act only", ("w", "act"))]
        if len(sys.argv) > 2 and sys.argv[2] == "groups":
            runs = [("w+act in %s only" % g, ("w", "act", g)) for g in ("pyr", "dec6", "dec5", "dec4", "dec3", "dec2", "ctx")]
        for name, on in runs:
            c = cfg_of(on)
            if len(on) == 3 and on[2] in ("pyr", "dec6", "dec5", "dec4", "dec3", "dec2", "ctx"):
                c["groups"] = (on[2],)
            out = forward(sd, x, c, all_levels=True)
            print("%-50s EPE flow2 %.3e | " % (name, O.epe(out[0], ref[0])) +
                  " ".join("L%d %.2e" % (l, O.epe(o, r)) for l, o, r in zip((3, 4, 5, 6), out[1:], ref[1:])), flush=True)


if __name__ == "__main__":
    main()
