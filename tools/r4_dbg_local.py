import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg, oracle_lib as O
mod = _pkg.load()
rows, mode = int(sys.argv[1]), sys.argv[2]
m = mod.generate_matrix(rows, 1024, 20, "gamma", 9)
nq = 40
xs = np.stack([mod.create_sample_vector(1024, True, False, True, 9100 + i) for i in range(nq)])
xs[7] = 0.0; xs[8] = -xs[8]; xs[20:24] *= np.float32(1e-2)
dxs = torch.from_numpy(xs).cuda()
res = {}
for name, env in (("exact", "0"), ("local", mode)):
    os.environ["TKSPMV_LOCAL"] = env
    eng = mod.SpMV(m.row, m.col, m.val, m.rows, m.cols, k=100, device=0)
    oi = torch.full((nq, 100), -1, dtype=torch.int32, device="cuda"); ov = torch.full((nq, 100), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.enqueue_batch(dxs.data_ptr(), nq, oi.data_ptr(), ov.data_ptr()); eng.synchronize()
    res[name] = (oi.cpu().numpy().view(np.uint32), ov.cpu().numpy(), eng.debug_counters(), eng.info())
    eng.close()
info = res["local"][3]
C = info["packet_entries"] // 64
packed = mod.Packed(m, k=100, nnz_per_lane=C, n_wave_partitions=(info["batch_mode"] >> 16) or info["n_wave_partitions"])
raw = packed.raw()
print("counters local:", res["local"][2])
for q in range(nq):
    yp, present = O.packed_scores(raw, xs[q], m.rows, C)
    ei, ev = O.select_topk(yp, present, 100)
    for name in ("exact", "local"):
        i_, v_ = res[name][0][q], res[name][1][q]
        if not (np.array_equal(i_, ei) and np.array_equal(v_.view(np.uint32), ev.view(np.uint32))):
            miss = sorted(set(ei.tolist()) - set(i_.tolist())); extra = sorted(set(i_.tolist()) - set(ei.tolist()))
            print(f"q={q} {name}: WRONG; missing {miss[:6]} (scores {[float(yp[r]) for r in miss[:6]]}) extra {extra[:6]}; kth {float(ev[-1])}; first diff pos {int(np.argmax(i_ != ei))}")
print("done")
