"""diagnostic: per-tensor gradient error of the build vs the fp64 oracle, beside the fp32 reference's own"""
import sys, torch
sys.path.insert(0, ".")  # run from the repository root
from tests.gpu_util import model_from_golden
from tests.helpers import Golden, golden_loss, fp64_truth, rel_err
for name in sys.argv[1:]:
    g = Golden(name); m = model_from_golden(g)
    kw, target, wv = g.model_kwargs(device="cuda")
    loc, vloc = m(**kw)
    golden_loss(loc, vloc, target, wv).backward()
    G = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    _, _, tG, _ = fp64_truth(g)
    rows = []
    for k in g.gp:
        e_ref, e_got = rel_err(g.gp[k], tG[k]), rel_err(G[k], tG[k])
        rows.append((e_got / max(e_ref, 1e-12), k, e_got, e_ref, float(tG[k].abs().max())))
    rows.sort(reverse=True)
    print(name, "worst ratios (got/ref):")
    for r in rows[:6]: print("  %-42s got %.2e ref %.2e ratio %5.1f  max|g| %.2e" % (r[1], r[2], r[3], r[0], r[4]))
