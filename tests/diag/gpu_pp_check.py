"""A/B a GEMM route inside the bf16 path: `save FILE` stores the network output and every debug tap of one
forward pass (C2, B=8, L=16384); `cmp FILE1 FILE2` prints the relative difference per tap.
Run `save` twice in separate processes with different ADF_GEMM_* environment variables."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch


def save(path):
    import audiodiffuser_amd as A
    from audiodiffuser_amd.weights import generate_noise
    from gpu_helpers import make_net
    cfg = A.config_c2()
    B = int(os.environ.get("B", "8"))
    x = generate_noise(0, B, 16384) * 0.7
    t = torch.linspace(-1.0, 0.5, B)
    kw = {}
    if os.environ.get("CLASSES"):        # class-conditional variant of the net: labels per sample (FiLM gets the class addend)
        cfg.class_cond, cfg.num_classes = True, 10
        kw = {"classes": (torch.arange(B) % 10).cuda(), "cond_drop_prob": 0.0}
    net, w = make_net(cfg, "bf16", int(os.environ.get("FLAGS", "0")))
    y = net(x.cuda(), t.cuda(), **kw)
    torch.cuda.synchronize()
    hd = net.native(torch.device("cuda", 0))
    out = {"out": y.float().cpu()}
    for name in hd.tap_names():
        out[name] = hd.tap(name, B, y.device).float().cpu()
    torch.save(out, path)
    print("saved", len(out), "tensors; out finite:", bool(torch.isfinite(out["out"]).all()), flush=True)


def cmp(p1, p2):
    a, b = torch.load(p1), torch.load(p2)
    worst = (0.0, "")
    for k in a:
        d = (a[k] - b[k]).norm() / (b[k].norm() + 1e-30)
        m = (a[k] - b[k]).abs().max()
        worst = max(worst, (float(d) if float(d) == float(d) else 1e30, k))
        if os.environ.get("VERBOSE") or not (float(d) <= 2e-2):
            nbad = int((~torch.isfinite(a[k])).sum())
            print(f"{k:40s} {tuple(a[k].shape)} rel {float(d):.3e} maxabs {float(m):.3e} nonfinite {nbad}")
            if os.environ.get("PERSAMPLE"):
                for bi in range(a[k].shape[0]):
                    da = a[k][bi] - b[k][bi]
                    fin = torch.isfinite(da)
                    print("    sample", bi, "nonfinite", int((~fin).sum()), "rel(finite)", float(da[fin].norm() / (b[k][bi][fin].norm() + 1e-30)))
    print("worst", worst, "out rel", float((a['out'] - b['out']).norm() / b['out'].norm()))


if __name__ == "__main__":
    if sys.argv[1] == "save":
        save(sys.argv[2])
    else:
        cmp(sys.argv[2], sys.argv[3])
