"""diagnostic: error levels of the bf16 operand mode -- HIP(bf16) against its mirror (oracle/factored.py, Config.bf16) in
fp32 and fp64, and against the fp32 reference goldens; input for the tolerances of tests/test_gpu_bf16.py.
    python tools/gpu_bf16err.py c16_two_graphs ragged3_gravity ..."""
import dataclasses, sys, torch
sys.path.insert(0, ".")
import fastegnn_amd
from oracle import factored as F
from tests.helpers import Golden, golden_loss, rel_err


def mirror(g, dt):
    cfg = dataclasses.replace(g.cfg, bf16=True)
    p = g.tensors(g.params, dtype=dt); kw, target, wv = g.model_kwargs(dtype=dt)
    loc, vloc, ctx = F.model_forward(p, cfg, **kw)
    l2 = loc.clone().requires_grad_(True); v2 = vloc.clone().requires_grad_(True)
    golden_loss(l2, v2, target, wv).backward()
    G, gin = F.model_backward(p, cfg, ctx, l2.grad, v2.grad)
    return loc, vloc, G


for name in sys.argv[1:]:
    g = Golden(name); c = g.cfg
    m = fastegnn_amd.FastEGNN(c.node_feat_nf, c.node_attr_nf, c.edge_attr_nf, c.hidden_nf, c.virtual_channels, device="cuda",
                              n_layers=c.n_layers, residual=c.residual, attention=c.attention, normalize=c.normalize, tanh=c.tanh,
                              gravity=c.gravity, mlp_dtype=torch.bfloat16)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in g.params.items()}, strict=True); m = m.cuda()
    kw, target, wv = g.model_kwargs(device="cuda")
    loc, vloc = m(**kw)
    golden_loss(loc, vloc, target, wv).backward()
    G = {k: (p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in m.named_parameters()}
    l32, v32, G32 = mirror(g, torch.float32)
    l64, v64, G64 = mirror(g, torch.float64)
    x0 = torch.from_numpy(g.inp["node_loc"]).double()
    print(f"== {name}: loc vs mirror32 {rel_err(loc, l32):.1e} mirror64 {rel_err(loc, l64):.1e} ref {rel_err(loc, g.out['loc']):.1e} | "
          f"disp vs mirror64 {rel_err(loc.cpu().double() - x0, l64 - x0):.1e} (mirror32: {rel_err(l32.double() - x0, l64 - x0):.1e}) "
          f"ref {rel_err(loc.cpu().double() - x0, torch.from_numpy(g.out['loc']).double() - x0):.1e} | vloc mirror64 {rel_err(vloc, v64):.1e}")
    rows = []
    for k in g.gp:
        if float(abs(torch.as_tensor(g.gp[k])).max()) == 0: continue
        rows.append((rel_err(G[k], G64[k]), rel_err(G32[k], G64[k]), rel_err(G[k], g.gp[k]), k))
    rows.sort(reverse=True)
    for r in rows[:5]: print("   %-44s got-vs-mirror64 %.1e  mirror32-vs-64 %.1e  got-vs-fp32ref %.1e" % (r[3], r[0], r[1], r[2]))
    import statistics
    print("   median got-vs-mirror64 %.1e, median mirror32-vs-64 %.1e, median/max vs fp32 ref %.1e / %.1e" % (
        statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows), statistics.median(r[2] for r in rows), max(r[2] for r in rows)))
