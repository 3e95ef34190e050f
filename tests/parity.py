"""Set-based parity metrics between oracle output and HIP output.

The reference's own notion of "correct" is sorted-set equality (testScripts/testOxfordDataset.sh.in:
140-154): the order of features is the order of atomicAdd and is not a contract.
"""
import numpy as np
from scipy.spatial import cKDTree

# north_star tolerances: coordinates / sigma 1e-3, descriptor L2 distance 1e-3
TOL_POS = 1e-3
TOL_SIGMA_REL = 1e-3
TOL_ORI = 1e-3
TOL_DESC = 1e-3


def _ang_diff(a, b):
    d = np.abs(a - b) % (2 * np.pi)
    return np.minimum(d, 2 * np.pi - d)


def match_features(fa, da, fb, db, norm_scale=1.0):
    """fa/fb: structured feature arrays (debug_octave,xpos,ypos,sigma,num_ori,orientation,desc_idx);
    da/db: (n,128) descriptor arrays.  Returns a dict of parity fractions (of set A)."""
    res = {"n_a": len(fa), "n_b": len(fb)}
    if len(fa) == 0 or len(fb) == 0:
        res.update(kp_match=float(len(fa) == len(fb)), ori_match=1.0, desc_match=1.0, max_desc_dist=0.0)
        return res
    # nearest neighbour in (x, y, octave*1e4) -- different octaves never match
    ka = np.stack([fa["xpos"], fa["ypos"], fa["debug_octave"] * 1e4, np.log2(fa["sigma"]) * 50.0], 1)
    kb = np.stack([fb["xpos"], fb["ypos"], fb["debug_octave"] * 1e4, np.log2(fb["sigma"]) * 50.0], 1)
    tree = cKDTree(kb)
    kq = min(4, len(fb))
    _, idx_all = tree.query(ka, k=kq)
    idx_all = idx_all.reshape(len(fa), kq)
    kp_ok = np.zeros(len(fa), bool)
    ori_ok = np.zeros(len(fa), bool)
    n_desc = 0
    n_desc_ok = 0
    max_dd = 0.0
    used = set()
    for i in range(len(fa)):
        a = fa[i]
        tol = TOL_POS * max(1.0, float(a["sigma"]))
        j = -1
        # the reference algorithm can emit exact duplicates (two start pixels refined to the same
        # position), so take the nearest *unused* candidate within tolerance
        for cand in idx_all[i]:
            cand = int(cand)
            b = fb[cand]
            if cand in used:
                continue
            if (abs(a["xpos"] - b["xpos"]) <= tol and abs(a["ypos"] - b["ypos"]) <= tol
                    and abs(a["sigma"] - b["sigma"]) <= TOL_SIGMA_REL * a["sigma"]):
                j = cand
                break
        if j < 0:
            continue
        b = fb[j]
        used.add(j)
        kp_ok[i] = True
        na, nb = int(a["num_ori"]), int(b["num_ori"])
        if na != nb:
            continue
        oa = a["orientation"][:na]
        ob = b["orientation"][:nb]
        # match orientations by nearest angle
        good = True
        pairs = []
        taken = set()
        for p in range(na):
            dd = _ang_diff(oa[p], ob)
            for q in taken:
                dd[q] = np.inf
            q = int(np.argmin(dd))
            if dd[q] > TOL_ORI:
                good = False
                break
            taken.add(q)
            pairs.append((p, q))
        if not good:
            continue
        ori_ok[i] = True
        for p, q in pairs:
            ia, ib = int(a["desc_idx"][p]), int(b["desc_idx"][q])
            if ia < 0 or ib < 0:
                continue
            n_desc += 1
            d = float(np.linalg.norm(da[ia].astype(np.float64) - db[ib].astype(np.float64))) / norm_scale
            max_dd = max(max_dd, d)
            if d <= TOL_DESC:
                n_desc_ok += 1
    res["kp_match"] = float(kp_ok.mean())
    res["ori_match"] = float(ori_ok.mean())
    res["desc_match"] = float(n_desc_ok / n_desc) if n_desc else 1.0
    res["desc_compared"] = n_desc
    res["max_desc_dist"] = max_dd
    return res


def sort_iext(a):
    """Canonical order for initial extrema: (lpos, ypos, xpos)."""
    order = np.lexsort((a["xpos"], a["ypos"], a["lpos"]))
    return a[order]
