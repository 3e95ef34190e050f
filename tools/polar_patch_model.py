"""dev tool (GPU box): upper bound of what a per-keypoint polar-gradient patch (one (|g|, theta) patch in LDS shared by
k_orientation and the keypoint's 1..4 descriptors; round-5 review, task 3) could save.  Two libraries are timed on the bench frame:
the product and a measurement build (-DPSX_MODEL_NOGRAD, popsift_amd/lib_model: tools/build_model_lib.sh) in which both kernels
take magnitude and angle of a pixel pair from ONE 8-byte read instead of five loads + hypot + atan2 (results are garbage; the
instruction stream is what a patch consumer would execute).  The difference is T_grad, the time the gradients cost today; a
patch computes them once per keypoint over the union of its windows instead of once per orientation pass + once per descriptor.
  python tools/polar_patch_model.py          (spawns itself once per library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure():
    import numpy as np
    from popsift_amd import capi
    from popsift_amd.synth import synth
    ctx = capi.Context(capi.default_config(octaves=5, sift_mode=2))
    ctx.upload(synth(1920, 1080, 1000))
    ctx.enable_timers(True)
    rows = []
    for i in range(45):
        ctx.extract(); ctx.sync()
        if i >= 5:
            rows.append(ctx.stage_times())
    med = [float(v) for v in np.median(np.array(rows), axis=0)]
    print(json.dumps({"counts": list(ctx.counts()), "stage_ms": med}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        measure()
        sys.exit(0)
    res = {}
    for name, lib in (("product", None), ("nograd", os.path.join(ROOT, "popsift_amd", "lib_model", "libpopsift_hip.so"))):
        env = dict(os.environ)
        if lib:
            env["POPSIFT_HIP_LIB"] = lib
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], capture_output=True, text=True, env=env)
        res[name] = json.loads(p.stdout.strip().splitlines()[-1])
    a, b = res["product"]["stage_ms"], res["nograd"]["stage_ms"]
    n_kp, n_desc = res["product"]["counts"][0], res["product"]["counts"][1]
    t_ori, t_desc = a[2] - b[2], a[3] - b[3]
    # the union of a keypoint's windows against what its passes visit today: the descriptor window is a rotated square of side
    # 12 sigma' (area 144) inside the disc of radius 8.5 sigma' (area 227) that holds it for every orientation; today's kernel
    # visits 144 / 0.67 (67 % useful lanes); the orientation window (radius 4.5 sigma') lies inside
    share = (n_kp * 227.0) / (n_desc * 144.0 / 0.67)
    saved = t_ori + t_desc * max(0.0, 1.0 - share)
    print("stage ms (pyramid, extrema, orientation + scan, descriptors): product %s   no-gradient build %s" % ([round(v, 4) for v in a], [round(v, 4) for v in b]))
    print("keypoints %d, descriptors %d (%.2f per keypoint)" % (n_kp, n_desc, n_desc / max(n_kp, 1)))
    print("T_grad: orientation %.4f ms, descriptors %.4f ms" % (t_ori, t_desc))
    print("a shared patch pays the descriptor's gradients once per keypoint over the disc that holds every orientation's window: "
          "%.2f of today's descriptor-side gradient work; upper bound of the saving %.4f ms of a %.4f ms frame = %.1f %%" % (
              share, saved, sum(a), 100.0 * saved / sum(a)))
    print("(upper bound: the patch's accurate-angle requirement of the orientation histogram, its LDS traffic and the keypoint-granular "
          "work distribution are not charged)")
