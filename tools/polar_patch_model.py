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
    libs = (("product", None), ("nograd", "lib_model"), ("noatomic", "lib_model_NOATOMIC"), ("nograd_noatomic", "lib_model_BOTH"))
    for name, sub in libs:
        env = dict(os.environ)
        if sub:
            lib = os.path.join(ROOT, "popsift_amd", sub, "libpopsift_hip.so")
            if not os.path.exists(lib):
                continue
            env["POPSIFT_HIP_LIB"] = lib
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], capture_output=True, text=True, env=env)
        res[name] = json.loads(p.stdout.strip().splitlines()[-1])
    prod = res["product"]
    n_kp, n_desc = prod["counts"]
    print("1080p bench frame, one context, medians of 40 frames (HIP event timers); the no-gradient builds produce other orientations,")
    print("hence other descriptor counts: the descriptor stage is compared PER DESCRIPTOR, the orientation stage per keypoint")
    for name, r in res.items():
        a = r["stage_ms"]
        print("  %-16s keypoints %6d descriptors %6d  orientation + scan %.4f ms (%.2f ns / keypoint)  descriptors %.4f ms (%.2f ns / descriptor)" % (
            name, r["counts"][0], r["counts"][1], a[2], a[2] * 1e6 / r["counts"][0], a[3], a[3] * 1e6 / r["counts"][1]))
    if "nograd" in res:
        ng = res["nograd"]
        t_ori = prod["stage_ms"][2] - ng["stage_ms"][2]
        t_desc = (prod["stage_ms"][3] / n_desc - ng["stage_ms"][3] / ng["counts"][1]) * n_desc
        # the union of a keypoint's windows against what its passes visit today: the descriptor window is a rotated square of
        # side 12 sigma' (area 144) inside the disc of radius 8.5 sigma' (area 227) that holds it for every orientation; today's
        # kernel visits 144 / 0.67 (67 % useful lanes); the orientation window (radius 4.5 sigma') lies inside
        share = (n_kp * 227.0) / (n_desc * 144.0 / 0.67)
        saved = max(t_ori, 0.0) + max(t_desc, 0.0) * max(0.0, 1.0 - share)
        frame = sum(prod["stage_ms"])
        print("T_grad (what hypot + atan2 + four of the five loads cost today): orientation %.4f ms, descriptors %.4f ms of a %.4f ms frame" % (t_ori, t_desc, frame))
        print("a shared patch computes the descriptor-side gradients once per keypoint over the disc that holds every orientation's window = %.2f of "
              "today's descriptor-side gradient work: upper bound of its saving %.4f ms = %.1f %% of the frame" % (share, saved, 100.0 * saved / frame))
        print("(upper bound: the accurate angle the orientation histogram needs, the patch's LDS traffic and keypoint-granular work distribution are not charged)")
    if "nograd_noatomic" in res:
        b = res["nograd_noatomic"]
        print("k_descriptors with NEITHER gradients NOR histogram atomics: %.2f ns per descriptor against %.2f -- what remains (window walk, weights, "
              "bin arithmetic, per-descriptor prologue / normalisation, load latency of the one remaining read) is %.0f %% of the kernel" % (
                  b["stage_ms"][3] * 1e6 / b["counts"][1], prod["stage_ms"][3] * 1e6 / n_desc,
                  100.0 * (b["stage_ms"][3] / b["counts"][1]) / (prod["stage_ms"][3] / n_desc)))
