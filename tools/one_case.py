"""dev tool (GPU box): one configuration against the oracle, printing every orientation / descriptor mismatch.
usage: python tools/one_case.py w h seed is_float '<json kw>'"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as oracle
from popsift_amd import capi
from popsift_amd.synth import synth, synth_float
from tests.parity import match_features
w, h, seed, is_float = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1"
kw = json.loads(sys.argv[5])
img = synth_float(w, h, seed) if is_float else synth(w, h, seed)
ref = oracle.run(oracle.default_config(**kw), img)
ctx = capi.Context(capi.default_config(**kw)); ctx.upload(img); ctx.extract()
fb, db = ctx.download()
fa, da = ref.features(), ref.descriptors()
m = match_features(fa, da, fb, db, norm_scale=float(2 ** kw.get("norm_multi", 0)))
print({k: m[k] for k in ("n_a", "kp_match", "ori_match", "desc_match", "max_desc_dist", "ori_miss", "desc_miss")})
for x in m["misses"][:10]:
    print("miss", x)
ea, eb = ref.extrema(), ctx.dump_extrema()
ka = np.lexsort((ea["ypos"], ea["xpos"], ea["octave"])); kb = np.lexsort((eb["ypos"], eb["xpos"], eb["octave"]))
ea, eb = ea[ka], eb[kb]
d = np.abs(ea["orientation"] - eb["orientation"]).max(axis=1)
bad = np.nonzero((d > 1e-4) | (ea["num_ori"] != eb["num_ori"]))[0]
for i in bad[:10]:
    print("extremum", i, "xy", float(ea["xpos"][i]), float(ea["ypos"][i]), "sigma", float(ea["sigma"][i]), "oracle", ea["num_ori"][i], ea["orientation"][i], "hip", eb["num_ori"][i], eb["orientation"][i])
