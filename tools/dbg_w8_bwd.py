import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import tests.test_ops_gpu as tg
from tests.test_ops_gpu import *
def run(T, B, ext):
    cellname, cell, dtype, H = "GRU", hl.GRU, hl.BF16, 256
    rng, G, U, W, b, h0, c0 = tg._rnn_problem(cellname, H, T, B, seed=11 + H)
    GH = G * H; td = ops.torch_dtype(dtype)
    rnd = (lambda a: host(dev(a, td)))
    xp = rng.standard_normal((T, B, GH)) * 0.5
    hs_o, cs_o, acts_o = vo.rnn_forward(cellname, xp, U, h0, None)
    hs_o, acts_o = rnd(hs_o), rnd(acts_o)
    dext = rnd(rng.standard_normal((T, B, H)) * 0.1) if ext else None
    dlast = rng.standard_normal((B, H)) * 0.1
    da_o, dU_o, dh0_o, dc0_o = vo.rnn_backward(cellname, hs_o, None, acts_o, U, dext, dlast)
    ut = ops.pack_recurrent(dev(U), cell, dtype, 1)
    lay = hl.TILE16Q
    da = torch.zeros((T, B, GH), dtype=td, device=DEV); rh = torch.zeros((T, B, H), dtype=td, device=DEV)
    dh0 = torch.zeros((B, H), device=DEV)
    acts_d = tile16(dev(acts_o, td), T * B, GH, True, paired="q")
    dext_d = tile16(dev(dext, td), T * B, H, True) if ext else None
    ops.rnn_bwd(cell, dtype, T, B, H, ut, dev(hs_o, td), None, acts_d, da, dhs_ext=dext_d, dh_last=dev(dlast), rh=rh, dh0=dh0, seq_layout=lay)
    torch.cuda.synchronize()
    e = np.abs(host(da) - da_o)
    print("T=%d B=%d ext=%d: da err max %.3e; per step max:" % (T, B, ext, e.max()), np.round(e.reshape(T, -1).max(1), 4))
    for g, nm in enumerate("zrc"):
        eg = e[:, :, g * H:(g + 1) * H]
        print("  gate", nm, "max %.3e" % eg.max(), "per tile:", np.round(eg.reshape(T, B, 16, 16).max((0, 1, 3)), 3))
    print("  dh0 err %.3e" % np.abs(host(dh0) - dh0_o).max(), " rh err %.3e" % np.abs(host(rh) - acts_o[:, :, H:2 * H] * hs_o[:-1]).max())
run(1, 16, False); run(1, 16, True); run(2, 16, False); run(8, 32, True)
