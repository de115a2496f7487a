import numpy as np, torch, sys
sys.path.insert(0,'.')
import mirror_nerf_amd as M
from oracle import mirror_nerf_oracle as O
rs = np.random.RandomState(11)
N, S, NI = 300, 64, 128
z = np.sort(rs.uniform(0.05, 8, (N, S)).astype(np.float32), 1)
w = (rs.uniform(0, 1, (N, S)) ** 8).astype(np.float32)
w[5] = 0; w[6, 10] = 1.0
mid = 0.5 * (z[:, :-1] + z[:, 1:])
smp = O.sample_pdf(mid, w[:, 1:-1], NI, det=True)
want = np.sort(np.concatenate([z, smp], -1), -1)
got = M.sample_pdf(torch.from_numpy(z).cuda(), torch.from_numpy(w).cuda(), NI, det=True).cpu().numpy()
d = np.abs(got-want)
bad = np.nonzero(d.max(1) > 2e-5)[0]
print("bad rays", bad[:20], len(bad))
for r in bad[:3]:
    j = np.nonzero(d[r] > 2e-5)[0]
    print(r, j[:10], got[r][j[:10]], want[r][j[:10]])
    # which samples of oracle are missing
    print(' oracle samples not in got:', [x for x in smp[r] if np.min(np.abs(got[r]-x))>1e-5][:5])
    print(' got values not in oracle:', [x for x in got[r] if np.min(np.abs(want[r]-x))>1e-5][:5])
