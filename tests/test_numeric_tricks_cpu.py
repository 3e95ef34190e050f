"""The arithmetic identities the kernels rely on where they replace an IEEE operation of the reference by something
cheaper, checked on the CPU (numpy float32; an fma is emulated in float64 wherever the exact product fits 53 bits, which
it does for every case below).  Each test names the kernel code it backs; the kernels themselves are held to the oracle
bit for bit by the GPU tests.
"""
import numpy as np

f32 = np.float32


def _fma(a, b, c):
    """fl32(a * b + c) with one rounding (operands float32; exact in float64 for the magnitudes used here)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def test_u8_over_255_without_a_division():
    """pyramid.hip l0_unorm8 / pyramid_alt.hip a_texel: q * RN(1/255) and one Newton correction is fl(q / 255) for all
    256 byte values (cudaReadModeNormalizedFloat, s_image.cu:147)."""
    q = np.arange(256, dtype=f32)
    c = f32(1.0) / f32(255.0)
    r = (q * c).astype(f32)
    e = _fma(np.full(256, -255.0, f32), r, q)
    got = _fma(e, np.full(256, c, f32), r)
    assert np.array_equal(got, (q / f32(255.0)).astype(f32))


def test_tap_offset_over_width_by_reciprocal_and_one_newton_step():
    """pyramid_alt.hip k_alt_h_input: float(offset) / W (s_pyramid_build_ra.cu:44) as k * RN(1/W) corrected once equals
    the IEEE quotient for every tap index up to 32 and every width below 2^17 (the kernel divides beyond that)."""
    W = np.arange(1, 1 << 17, dtype=np.float64)
    Wf = W.astype(f32)
    y = (f32(1.0) / Wf).astype(f32)
    for k in range(1, 33):
        kf = np.full(W.shape, k, f32)
        q0 = (kf * y).astype(f32)
        r = _fma(-Wf, q0, kf)
        q = _fma(r, y, q0)
        assert np.array_equal(q, (kf / Wf).astype(f32)), k


def test_division_by_three_in_three_instructions():
    """orient_desc.hip div3 (the six smoothing passes of the orientation histogram, s_orientation.cu:166-174): x * RN(1/3),
    the exact remainder in an fma, one correction -- the IEEE quotient on random floats over nine decades and on the
    integers a histogram can hold exactly."""
    rng = np.random.default_rng(5)
    x = np.concatenate([(rng.random(2_000_000) * 10.0 ** rng.integers(-4, 5, 2_000_000)).astype(f32),
                        np.arange(0, 1 << 20, dtype=f32)])
    y = f32(0.3333333432674407958984375)
    q = (x * y).astype(f32)
    r = _fma(np.full(x.shape, -3.0, f32), q, x)
    got = _fma(r, np.full(x.shape, y, f32), q)
    assert np.array_equal(got, (x / f32(3.0)).astype(f32))


def test_bin_pair_slot_is_a_three_bit_rotation():
    """orient_desc.hip bin_slot: the u64 that holds bins (fo, fo + 1) of a descriptor tile lives at word fo for even fo and
    at word 8 + fo - 1 for odd fo (two views of the 8 bins); as a byte offset that is the bin index rotated right by one
    bit, times 8 -- for any integer fo, negative ones included (two's complement & 7)."""
    for fo in range(-16, 17):
        h = fo & 7
        word = h if h % 2 == 0 else 8 + h - 1
        rot = ((((h << 3) | h) >> 1) & 7) << 3
        assert rot == word * 4, fo


def test_bounding_box_floor_of_the_extreme_corner():
    """orient_desc.hip k_descriptors: min over the corners of floor(p - b) == floor(min(p) - b) (and max / + likewise),
    because x -> floor(fl(x - b)) is monotone; the reference takes the floor per corner (s_desc_loop.cu:46-58)."""
    rng = np.random.default_rng(9)
    p = (rng.random((200_000, 4)) * 4000.0 - 100.0).astype(f32)
    p[: 50_000] = np.rint(p[: 50_000])                      # corners on integers and ties
    b = (rng.random(200_000) * 60.0).astype(f32)
    lo_each = np.floor((p - b[:, None]).astype(f32)).min(axis=1)
    lo_once = np.floor((p.min(axis=1) - b).astype(f32))
    hi_each = np.floor((p + b[:, None]).astype(f32)).max(axis=1)
    hi_once = np.floor((p.max(axis=1) + b).astype(f32))
    assert np.array_equal(lo_each, lo_once) and np.array_equal(hi_each, hi_once)


def test_one_minus_w_and_w_from_one_fma():
    """orient_desc.hip: the pair (1 - w, w) as fma(w, (-1, 1), (1, 0)) -- fl(1 - w) and w themselves, for every
    representable fraction the kernel can see (w = t - floor(t), t in (-12, 12))."""
    rng = np.random.default_rng(2)
    t = ((rng.random(1_000_000) * 24.0 - 12.0)).astype(f32)
    w = (t - np.floor(t)).astype(f32)
    lo = _fma(w, np.full(w.shape, -1.0, f32), np.full(w.shape, 1.0, f32))
    hi = _fma(w, np.full(w.shape, 1.0, f32), np.full(w.shape, 0.0, f32))
    assert np.array_equal(lo, (f32(1.0) - w).astype(f32)) and np.array_equal(hi, w)


def test_denormal_product_is_the_fixed_point_integer():
    """orient_desc.hip k_descriptors<DENORM>: with the scale 2^14 * 2^-149 folded into a weight, every product further down
    is a denormal float, i.e. a multiple of 2^-149 whose BIT PATTERN is that multiple -- the 18.14 fixed-point integer that
    ds_add_u64 adds, with no conversion.  Each multiplication rounds to that grid (nearest even): a product of a
    grid-rounded weight and a factor <= 1 is within one unit of the exact value (the kernel chains four: <= 2 units of
    2^-14 against sums of 10^2..10^3), and sums of such words are exact integer sums."""
    rng = np.random.default_rng(4)
    a = (rng.random(500_000) * 360.0).astype(f32)                      # magnitude x Gaussian weight
    b = rng.random(500_000).astype(f32)                                 # bin weight x tile weight, <= 1
    scaled = (a * f32(2.0) ** f32(-135)).astype(f32)                   # a on the 2^-14 grid (denormal: a < 512)
    prod = (scaled * b).astype(f32)                                     # rounds to a multiple of 2^-149 again
    bits = prod.view(np.uint32).astype(np.int64)
    assert (bits < (1 << 23)).all()                                     # still denormal: the exponent field is zero
    exact = a.astype(np.float64) * b.astype(np.float64) * 16384.0
    assert np.abs(bits - exact).max() <= 1.0
    # the word IS the value: bits * 2^-149 == prod, and adding the words adds the values
    assert np.array_equal(bits.astype(np.float64) * 2.0 ** -149, prod.astype(np.float64))
    s = (prod[:1000].astype(np.float64)).sum()
    assert bits[:1000].sum() * 2.0 ** -149 == s


def test_level0_row_form_needs_a_power_of_two_ratio():
    """pyramid.hip psx_level0_exact: (x + s)/W -+ k/W and (x -+ k + s)/W land on the same 1/256 sub-texel for every column
    and tap when the image / octave ratio is a power of two, and not in general (a fractional scale factor: some columns
    differ by one 1/256 step) -- the reason the kernels of the default path are gated on that ratio."""
    def positions(c, size):
        t = (c * f32(size)).astype(f32) - f32(0.5)
        fl = np.floor(t)
        return fl.astype(np.int64) * 256 + np.rint(((t - fl).astype(f32) * f32(256.0)).astype(f32)).astype(np.int64)

    def mismatches(w, W, shift, taps=8):
        x = np.arange(W, dtype=f32)
        bad = 0
        for k in range(1, taps + 1):
            for sg in (-1.0, 1.0):
                lit = (((x + f32(shift)) / f32(W)).astype(f32) + f32(sg) * (f32(k) / f32(W))).astype(f32)
                row = (((x + f32(sg * k)) + f32(shift)) / f32(W)).astype(f32)
                bad += int((positions(lit, w) != positions(row, w)).sum())
        return bad

    for w, W, s in [(333, 666, 1.0), (333, 666, 0.5), (1920, 3840, 1.0), (1920, 960, 0.5), (1920, 240, 0.5), (333, 333, 0.5), (251, 1004, 2.0)]:
        assert mismatches(w, W, s) == 0, (w, W, s)
    assert mismatches(333, 471, 0.5 * 2 ** 0.5) > 0 and mismatches(640, 906, 0.5 * 2 ** 0.5) > 0


def test_orientation_fast_bin_equals_exact_bin_outside_the_guard_band():
    """orient_desc.hip k_orientation: the histogram bin of a sample is roundf(36 (atan2f + pi) / 2 pi) with the angle rounded
    once from double (oracle/sift_oracle.c atan2f_1r).  The kernel evaluates that only within 2e-4 bins of a bin boundary and
    otherwise takes a degree-13 polynomial in bin units with v_rcp_f32 (1 ulp); this replays both in float32 on a million
    gradients (random, near-diagonal, near-axis, tiny, integer-valued like the ones of smooth 8-bit images) and checks that
    the fast bin is the exact bin wherever the guard does not send the sample to the exact path -- for a correctly rounded
    reciprocal and for one that is an ulp off either way."""
    rng = np.random.default_rng(11)
    n = 250_000
    g = rng.standard_normal((n, 2)) * 10.0 ** rng.integers(-3, 3, (n, 1))
    diag = rng.standard_normal((n, 1)) * 20.0
    near_diag = np.hstack([diag, diag * rng.choice([-1.0, 1.0], (n, 1))]) * (1.0 + rng.standard_normal((n, 2)) * 1e-6)
    near_axis = np.hstack([rng.standard_normal((n, 1)) * 30.0, rng.standard_normal((n, 1)) * 1e-4])
    ints = rng.integers(-40, 41, (n, 2)).astype(np.float64) * rng.choice([1.0, 0.5, 0.25], (n, 1))
    g = np.vstack([g, near_diag, near_axis, near_axis[:, ::-1], ints]).astype(f32)
    gx, gy = g[:, 0], g[:, 1]
    keep = (gx != 0) | (gy != 0)
    gx, gy = gx[keep], gy[keep]

    PI_F, PI2_F = f32(3.14159265358979323846), f32(6.28318530717958647692)
    exact_at = np.arctan2(gy.astype(np.float64), gx.astype(np.float64)).astype(f32)
    be = ((f32(36.0) * (exact_at + PI_F).astype(f32)).astype(f32) / PI2_F).astype(f32)
    exact_bin = (np.trunc(be) + (np.abs(be - np.trunc(be)) >= 0.5) * np.sign(be)).astype(np.int64)     # roundf: half away from zero
    exact_bin[exact_bin == 36] = 0

    KB = f32(5.729577951308232)
    coef = [(f32(c) * KB).astype(f32) for c in (0.006811792496591806, -0.0336042195558548, 0.07962366938591003,
                                                -0.1323334127664566, 0.19807815551757812, -0.3331736922264099, 0.9999961256980896)]
    ax, ay = np.abs(gx), np.abs(gy)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    den = np.maximum(mx, f32(1e-30))
    rc0 = (f32(1.0) / den).astype(f32)
    for ulp in (0, 1, -1):
        rc = rc0 if ulp == 0 else np.nextafter(rc0, np.where(ulp > 0, f32(np.inf), f32(0.0)).astype(f32)).astype(f32)
        a = (mn * rc).astype(f32)
        s2 = (a * a).astype(f32)
        r = np.full(a.shape, coef[0], f32)
        for c in coef[1:]:
            r = _fma(r, s2, np.full(a.shape, c, f32))
        r = (r * a).astype(f32)
        at = np.where(ay > ax, (f32(9.0) - r).astype(f32), r)
        at = np.where(gx < 0, (f32(18.0) - at).astype(f32), at)
        at = np.where(gy < 0, -at, at)
        bfast = (at + f32(18.0)).astype(f32)
        bfl = np.floor(bfast)
        frac = (bfast - bfl).astype(f32)
        fast_bin = bfl.astype(np.int64) + (frac >= 0.5)
        fast_bin[fast_bin == 36] = 0
        guard = np.abs((frac - f32(0.5)).astype(f32)) < f32(2e-4)
        wrong = (fast_bin != exact_bin) & ~guard
        assert not wrong.any(), (ulp, int(wrong.sum()), gx[wrong][:4], gy[wrong][:4])
        assert guard.mean() < 0.35                 # (the near-diagonal / integer families sit on boundaries by construction)
        # distance of the fast value from the exact expression: the kernel's comment budgets < 1.2e-5 bins
        d = np.abs(bfast.astype(np.float64) - be.astype(np.float64))
        d = np.minimum(d, 36.0 - d)
        assert d.max() < 1.5e-5, d.max()


def test_atan_polynomial_error_bound():
    """orient_desc.hip fast_atan2 / the packed atan of k_descriptors and k_orientation: the degree-13 odd polynomial on
    [0, 1], evaluated in float32 with fmas as the kernels do, stays within 3.5e-7 rad of atan (the comments say 3.3e-7)."""
    a = np.linspace(0.0, 1.0, 2_000_001).astype(f32)
    s = (a * a).astype(f32)
    coef = [f32(c) for c in (0.006811792496591806, -0.0336042195558548, 0.07962366938591003, -0.1323334127664566,
                             0.19807815551757812, -0.3331736922264099, 0.9999961256980896)]
    r = np.full(a.shape, coef[0], f32)
    for c in coef[1:]:
        r = _fma(r, s, np.full(a.shape, c, f32))
    r = (r * a).astype(f32)
    err = np.abs(r.astype(np.float64) - np.arctan(a.astype(np.float64)))
    assert err.max() < 3.5e-7, err.max()


def test_row_split_of_the_orientation_window_by_reciprocal():
    """orient_desc.hip k_orientation: i / pw as (int)((i + 0.5) * rcp(pw)) with v_rcp_f32 (1 ulp): exact for every pair count
    a window can have (radius <= 3 * 1.5 * sigma_max ~ 40) and every index in it, reciprocal an ulp off either way included."""
    for pw in range(1, 48):
        i = np.arange(0, pw * 100, dtype=np.int64)
        base = f32(f32(1.0) / f32(pw))
        for r in (base, np.nextafter(f32(base), f32(np.inf)), np.nextafter(f32(base), f32(0.0))):
            q = ((i.astype(f32) + f32(0.5)) * f32(r)).astype(f32).astype(np.int64)
            assert np.array_equal(q, i // pw), (pw, float(r))


def test_x2_upsampling_has_constant_filter_weights():
    """pyramid.hip k_level0_x2: with W = 2 w the 1.8 fixed-point weight of output column X is 0 / 1/2 by parity for the
    sampling shift 1.0 (PopSift, VLFeat) and 3/4 / 1/4 for 0.5 (OpenCV), and the left texel is X/2 - (parity ? 0 : 1) resp.
    (X - 1) >> 1 -- for every width the kernel accepts; what the kernel's constants and its gate rest on."""
    for w in (4, 5, 7, 64, 333, 1920, 4096, 5000):
        W = 2 * w
        X = np.arange(-16, W + 16, dtype=f32)
        for shift in (1.0, 0.5):
            cn = ((X + f32(shift)) / f32(W)).astype(f32)
            tb = ((cn * f32(w)).astype(f32) - f32(0.5)).astype(f32)
            fl = np.floor(tb)
            al = (np.rint(((tb - fl).astype(f32) * f32(256.0)).astype(f32)) * f32(1.0 / 256.0)).astype(f32)
            i0 = fl.astype(np.int64)
            Xi = X.astype(np.int64)
            # a weight of 1 on the left neighbour's right texel is the same sample as a weight of 0 one texel further: compare
            # the sampled position i0 + al
            pos = i0 + al.astype(np.float64)
            want = Xi / 2.0 if shift == 1.0 else Xi / 2.0 - 0.25
            assert np.array_equal(pos, want), (w, shift)


def test_descriptor_row_spans_contain_every_window_pixel():
    """orient_desc.hip k_descriptors: a row of the rotated 5 x 5-SBP window is walked from xa to xb, computed from the two
    linear constraints with one pixel of slack (round 4; 2-3 before) and reciprocals from v_rcp_f32; the exact predicate
    |u - 1.5| < 2.5, |v - 1.5| < 2.5 decides inside.  Float32 replay for random keypoints: every pixel that passes the
    predicate lies inside its row's span (reciprocals exact and an ulp off either way) -- the spans only ever skip pixels
    that are outside the window, which is why tightening them left the descriptors bit-identical."""
    rng = np.random.default_rng(21)
    checked = 0
    for _ in range(1200):
        x = f32(rng.uniform(30, 400)); y = f32(rng.uniform(30, 300))
        ang = f32(rng.uniform(-np.pi, np.pi)) if rng.random() > 0.15 else f32(rng.choice([0.0, np.pi / 2, -np.pi / 2, np.pi / 4, np.pi]))
        SBP = f32(3.0 * rng.uniform(1.2, 6.7))
        W, H = 460, 360
        cos_t, sin_t = f32(np.cos(np.float64(ang))), f32(np.sin(np.float64(ang)))
        csbp, ssbp = f32(cos_t * SBP), f32(sin_t * SBP)
        crsbp, srsbp = f32(cos_t / SBP), f32(sin_t / SBP)
        bsz = f32(abs(csbp) + abs(ssbp))
        px = [f32(f32(csbp * ox) + f32(f32(-ssbp * oy) + x)) for ox in (-1.5, 1.5) for oy in (-1.5, 1.5)]
        py = [f32(f32(csbp * oy) + f32(f32(ssbp * ox) + y)) for ox in (-1.5, 1.5) for oy in (-1.5, 1.5)]
        xmin = max(1, int(np.floor(f32(min(px) - bsz)))); xmax = min(W - 2, int(np.floor(f32(max(px) + bsz))))
        ymin = max(1, int(np.floor(f32(min(py) - bsz)))); ymax = min(H - 2, int(np.floor(f32(max(py) + bsz))))
        if xmin > xmax or ymin > ymax:
            continue
        use_c, use_s = abs(crsbp) > 1e-6, abs(srsbp) > 1e-6
        jj = np.arange(xmin, xmax + 1, dtype=np.int64)
        dx = (jj.astype(f32) - x).astype(f32)
        for ulp in (0, 1, -1):
            def rcp(v):
                r = f32(f32(1.0) / v)
                return r if ulp == 0 else np.nextafter(r, f32(np.inf) if ulp > 0 else f32(-np.inf))
            rcc = rcp(crsbp) if use_c else f32(0.0)
            rcs = rcp(srsbp) if use_s else f32(0.0)
            for ii in range(ymin, ymax + 1):
                dyk = f32(f32(ii) - y)
                ub = f32(np.float64(srsbp) * np.float64(dyk) + 1.5)          # fmaf
                vb = f32(np.float64(crsbp) * np.float64(dyk) + 1.5)
                lo, hi = f32(xmin) - x, f32(xmax) - x
                if use_c:
                    t1, t2 = f32(f32(-1.0 - ub) * rcc), f32(f32(4.0 - ub) * rcc)
                    lo, hi = max(lo, min(t1, t2)), min(hi, max(t1, t2))
                if use_s:
                    t1, t2 = f32(f32(vb - 4.0) * rcs), f32(f32(vb + 1.0) * rcs)
                    lo, hi = max(lo, min(t1, t2)), min(hi, max(t1, t2))
                u = (np.float64(crsbp) * dx.astype(np.float64) + np.float64(ub)).astype(f32)      # pk_fma
                v = (np.float64(-srsbp) * dx.astype(np.float64) + np.float64(vb)).astype(f32)
                inside = (np.abs((u - f32(1.5)).astype(f32)) < 2.5) & (np.abs((v - f32(1.5)).astype(f32)) < 2.5)
                if not inside.any():
                    continue
                assert lo <= hi, (x, y, ang, SBP, ii)
                xa = max(xmin, int(np.floor(f32(x + lo)))) & ~1
                xb = min(xmax, int(np.floor(f32(x + hi))) + 1)
                first, last = int(jj[inside][0]), int(jj[inside][-1])
                assert xa <= first and last <= xb, (x, y, ang, SBP, ii, xa, xb, first, last)
                checked += 1
    assert checked > 100_000


def test_two_bin_plateau_peaks_on_the_common_boundary():
    """orient_desc.hip k_orientation: with the peak test hval > hp && hval >= hn a two-bin plateau (an exact tie that only
    the integer histogram can produce; the reference's float sums break it by rounding noise) is a peak at its LEFT bin,
    and the parabola through (hp, h, h) puts the refined position 1.5 bins right of the previous bin, i.e. on the boundary
    between the two equal bins -- where the reference's answer lies up to its noise.  Exactly one peak per plateau, none on
    a flat histogram, unchanged where neighbours differ."""
    def peaks(h):
        out = []
        n = len(h)
        for b in range(n):
            hp, hv, hn = h[(b - 1) % n], h[b], h[(b + 1) % n]
            if hv > hp and hv >= hn:
                num = 3.0 * hp - 4.0 * hv + hn
                den = 2.0 * (hp - 2.0 * hv + hn)
                nb = num / den
                if 0.0 <= nb <= 2.0:
                    out.append(((b - 1) % n) + nb)
        return out

    h = np.array([0.1, 0.4, 1.0, 1.0, 0.5, 0.2] + [0.05] * 30)
    assert len(peaks(h)) == 1 and abs(peaks(h)[0] - 2.5) < 1e-9   # the boundary between bins 2 and 3
    assert peaks(np.zeros(36)) == []
    h2 = np.array([0.1, 0.4, 1.0, 0.9, 0.5, 0.2] + [0.05] * 30)
    strict = [b for b in range(36) if h2[b] > max(h2[b - 1], h2[(b + 1) % 36])]
    assert strict == [2] and len(peaks(h2)) == 1 and 1.5 < peaks(h2)[0] < 2.5
    h3 = np.array([0.05] * 34 + [1.0, 1.0])                    # a plateau across the wrap-around neighbour
    pw = peaks(np.roll(h3, 1))
    assert len(pw) == 1 and abs(pw[0] - 35.5) < 1e-9


def test_matcher_prefilter_margin_holds():
    """match.hip k_match_mfma discards a pair when its approximate distance s = |l|^2 + |r|^2 - 2 <f16(2^k l), f16(2^k r)> / 4^k lies
    more than 2 E_l above a running second-smallest s, with E_l = 0.00197 |l| Rmax + 2e-7 M (|l| + Rmax) + 4e-5 (|l|^2 + Rmax^2)
    claimed to bound |s - d| for the float distance d of the reference's operation tree.  Replay in numpy -- float16
    conversion (round to nearest even), also with every f16 denormal flushed to zero as a matrix unit may do, exact products,
    float32 sums -- on the descriptor families of the GPU test: the worst |s - d| / E_l must stay below 1 (it sits near 0.3)."""
    rng = np.random.default_rng(17)

    def unit(n):
        v = rng.random((n, 128), dtype=f32) ** 4
        return np.sqrt(v / v.sum(1, keepdims=True)).astype(f32)

    def tree(l, r):
        """the reference's distance (features.cu:160-189): per float4 a left-to-right fma chain, then the 16/8/4/2/1 tree"""
        q = (l[:, None, :] - r[None, :, :]).astype(f32).reshape(len(l), len(r), 32, 4)
        p = (q[..., 1] * q[..., 1]).astype(f32)
        for c in (0, 2, 3):
            p = (q[..., c].astype(np.float64) * q[..., c].astype(np.float64) + p.astype(np.float64)).astype(f32)
        for step in (16, 8, 4, 2, 1):
            p = (p[..., :step] + p[..., step:2 * step]).astype(f32)
        return p[..., 0]

    families = [("random", rng.random((96, 128), dtype=f32), rng.random((160, 128), dtype=f32)),
                ("unit", unit(96), unit(160)),
                ("unit x 512", (unit(96) * f32(512)).astype(f32), (unit(160) * f32(512)).astype(f32)),
                ("unit x 1e-9", (unit(96) * f32(1e-9)).astype(f32), (unit(160) * f32(1e-9)).astype(f32)),
                ("unit x 1e9", (unit(96) * f32(1e9)).astype(f32), (unit(160) * f32(1e9)).astype(f32))]
    l, r = unit(96), unit(160)
    l[::3] *= f32(1e-4); r[::5] *= f32(1e-3); r[7] = l[4]; r[8] = l[4] + f32(1e-5)
    families.append(("mixed norms + near duplicates", l, r))
    sparse = unit(96); sparse[:, 16:] = 0; sparse = (sparse / np.linalg.norm(sparse, axis=1, keepdims=True)).astype(f32)
    families.append(("sparse", sparse, unit(160)))
    worst = 0.0
    for name, l, r in families:
        nl = (l.astype(np.float64) ** 2).sum(1).astype(f32); nr = (r.astype(np.float64) ** 2).sum(1).astype(f32)
        M = f32(np.sqrt(max(nl.max(), nr.max()))); rmax2 = nr.max(); rmax = f32(np.sqrt(rmax2))
        k = int(np.floor(np.log2(f32(16384.0) / M)))
        sc = f32(2.0) ** k
        d = tree(l, r).astype(np.float64)
        for flush in (False, True):
            lh, rh = (l * sc).astype(np.float16), (r * sc).astype(np.float16)
            assert np.isfinite(lh).all() and np.isfinite(rh).all()
            if flush:
                lh = np.where(np.abs(lh) < np.float16(2.0 ** -14), np.float16(0), lh)
                rh = np.where(np.abs(rh) < np.float16(2.0 ** -14), np.float16(0), rh)
            dot = (lh.astype(f32) @ rh.astype(f32).T).astype(f32)                     # exact products, float32 accumulation
            s = (nl[:, None].astype(np.float64) + (f32(-2.0) / (sc * sc)) * dot.astype(np.float64) + nr[None, :]).astype(f32)
            E = (f32(0.00197) * np.sqrt(nl) * rmax + f32(2e-7) * M * (np.sqrt(nl) + rmax) + f32(4e-5) * (nl + rmax2)).astype(np.float64)
            ratio = float((np.abs(s.astype(np.float64) - d) / E[:, None]).max())
            worst = max(worst, ratio)
            assert ratio < 1.0, (name, flush, ratio)
    print("matcher prefilter: worst |s - d| / E_l = %.3f" % worst)


def _alt_window_box(x, y, sbp):
    """orient_desc.hip k_descriptors_alt: the window of the plane a workgroup stages for one descriptor."""
    ei = int(np.ceil(_fma(np.asarray([3.5356], f32), np.asarray([sbp], f32), np.asarray([2.51], f32))[0]))
    return int(np.floor(x)) - ei, int(np.floor(y)) - ei, 2 * ei + 2


def _alt_reads_bilinear(px, py, cos_t, sin_t):
    """Integer texel coordinates alt_gradiant_rot touches for sample points (px, py): the four stencil taps, each the
    2 x 2 texels at floor((c + 0.5) - 0.5) and + 1.  Returns (min x, max x, min y, max y)."""
    xs, ys = [], []
    for dx, dy in ((cos_t, sin_t), (-cos_t, -sin_t), (-sin_t, cos_t), (sin_t, -cos_t)):
        tx = ((px + f32(dx)).astype(f32) + f32(0.5)).astype(f32) - f32(0.5)
        ty = ((py + f32(dy)).astype(f32) + f32(0.5)).astype(f32) - f32(0.5)
        xs.append(np.floor(tx)); ys.append(np.floor(ty))
    xs, ys = np.concatenate(xs), np.concatenate(ys)
    return xs.min(), xs.max() + 1, ys.min(), ys.max() + 1


def test_alt_descriptor_window_holds_every_texel_the_modes_read():
    """orient_desc.hip k_descriptors_alt stages texels floor(x) - Ei .. floor(x) + Ei + 1 (Ei = ceil(3.5356 SBP + 2.51)) per
    axis and its waves read the window WITHOUT a range check.  The sample coordinates of the four modes are re-evaluated
    here in float32 (the kernel's expressions: s_desc_igrid.cu / s_desc_notile.cu / s_desc_iloop.cu / s_desc_grid.cu as
    restated in alt_tiles) for random keypoints, sizes and orientations -- also at the plane's corner and for tiny and
    maximal SBP -- and every texel they touch must lie inside the window."""
    rng = np.random.default_rng(11)
    n_checked = 0
    for trial in range(400):
        sbp = f32(rng.choice([0.05, 0.7, 3.0, 5.4, 8.1, 10.8, 10.88]) if trial % 3 == 0 else rng.uniform(0.3, 10.88))
        ang = f32(rng.uniform(-np.pi, np.pi) if trial % 5 else rng.choice([0.0, np.pi / 4, np.pi / 2, -np.pi / 4, np.pi]))
        x = f32(rng.uniform(-0.5, 4000.0) if trial % 4 else rng.choice([0.0, 0.999, 1.0, 3839.5]))
        y = f32(rng.uniform(-0.5, 2200.0) if trial % 4 else rng.choice([0.0, 0.001, 2159.0]))
        cos_t, sin_t = f32(np.cos(np.float64(ang))), f32(np.sin(np.float64(ang)))
        bx0, by0, bw = _alt_window_box(x, y, sbp)
        assert bw <= 84, (sbp, bw)                        # SBP <= 10.88 = sigma <= 3.63 fits ALT_WIN_MAX

        def inside(lims, what):
            x0, x1, y0, y1 = lims
            assert bx0 <= x0 and x1 <= bx0 + bw - 1 and by0 <= y0 and y1 <= by0 + bw - 1, \
                (what, float(sbp), float(ang), float(x), float(y), lims, (bx0, by0, bw))

        # igrid / notile: the lattice step = -2.5 + 1/16 + k/8, k = 0..39, rotated, times SBP, around (x, y)
        k = np.arange(40, dtype=f32)
        step = (f32(-2.5) + f32(1.0 / 16.0) + k / f32(8.0)).astype(f32)
        sx, sy = np.meshgrid(step, step)
        sx, sy = sx.ravel(), sy.ravel()
        ptx = _fma(np.full(sx.shape, cos_t, f32), sx, (-sin_t * sy).astype(f32))
        pty = _fma(np.full(sx.shape, cos_t, f32), sy, (sin_t * sx).astype(f32))
        px = _fma(ptx, np.full(sx.shape, sbp, f32), np.full(sx.shape, x, f32))
        py = _fma(pty, np.full(sx.shape, sbp, f32), np.full(sx.shape, y, f32))
        inside(_alt_reads_bilinear(px, py, cos_t, sin_t), "igrid/notile")

        # iloop: tile centres +-0.5, +-1.5 (x SBP, rotated) + the 32 x 32 lattice over [-bsz, bsz)^2 x SBP, inside |n| < 1 only
        csbp, ssbp = f32(cos_t * sbp), f32(sin_t * sbp)
        bsz = f32(abs(cos_t) + abs(sin_t))
        sub = np.arange(32, dtype=f32)
        d = (-bsz + (sub * bsz).astype(f32) / f32(16.0)).astype(f32)
        dx, dy = np.meshgrid(d, d)
        dx, dy = dx.ravel(), dy.ravel()
        nx = _fma(np.full(dx.shape, cos_t, f32), dx, (sin_t * dy).astype(f32))
        ny = _fma(np.full(dx.shape, cos_t, f32), dy, (-sin_t * dx).astype(f32))
        m = (np.abs(nx) < 1) & (np.abs(ny) < 1)
        for offx in (-1.5, -0.5, 0.5, 1.5):
            for offy in (-1.5, -0.5, 0.5, 1.5):
                tpx = _fma(np.asarray([csbp], f32), np.asarray([offx], f32), np.asarray([-ssbp * f32(offy)], f32))[0]
                tpy = _fma(np.asarray([csbp], f32), np.asarray([offy], f32), np.asarray([ssbp * f32(offx)], f32))[0]
                jj = ((x + tpx).astype(f32) + (dx[m] * sbp).astype(f32)).astype(f32)
                ii = ((y + tpy).astype(f32) + (dy[m] * sbp).astype(f32)).astype(f32)
                inside(_alt_reads_bilinear(jj, ii, cos_t, sin_t), "iloop")

        # grid: 16 x 16 samples per tile snapped to pixels through roundf and an (int) conversion, point reads at +-1
        xd = np.arange(16, dtype=f32) + f32(0.5)
        gx, gy = np.meshgrid(xd, xd)
        gx, gy = gx.ravel(), gy.ravel()
        ldx, ldy = f32(-cos_t + sin_t), f32(-cos_t - sin_t)
        rsx, rsy, usx, usy = f32(cos_t / f32(8)), f32(sin_t / f32(8)), f32(-sin_t / f32(8)), f32(cos_t / f32(8))
        one = np.ones(gx.shape, f32)
        pox = _fma(gy, one * usx, _fma(gx, one * rsx, one * ldx))
        poy = _fma(gy, one * usy, _fma(gx, one * rsy, one * ldy))
        for offx in (-1.5, -0.5, 0.5, 1.5):
            for offy in (-1.5, -0.5, 0.5, 1.5):
                tpx = _fma(np.asarray([csbp], f32), np.asarray([offx], f32), _fma(np.asarray([-ssbp], f32), np.asarray([offy], f32), np.asarray([x], f32)))[0]
                tpy = _fma(np.asarray([csbp], f32), np.asarray([offy], f32), _fma(np.asarray([ssbp], f32), np.asarray([offx], f32), np.asarray([y], f32)))[0]
                rx = _fma(pox, one * sbp, one * tpx)
                ry = _fma(poy, one * sbp, one * tpy)
                rnd = lambda v: (np.sign(v) * np.floor(np.abs(v) + f32(0.5))).astype(f32)          # roundf: half away from zero
                pix_x, pix_y = (rnd(rx) - tpx).astype(f32), (rnd(ry) - tpy).astype(f32)
                ix = np.trunc((tpx + pix_x).astype(f32))
                iy = np.trunc((tpy + pix_y).astype(f32))
                inside((ix.min() - 1, ix.max() + 1, iy.min() - 1, iy.max() + 1), "grid")
        n_checked += 1
    assert n_checked == 400


def _tex_axis(tb):
    """the texture unit on one axis: texel index floor(tb) and the 1.8 fixed-point weight (pyramid_alt.hip a_axis / plane_linear_1d)"""
    ft = np.floor(tb).astype(f32)
    w = (np.rint(((tb - ft).astype(f32) * f32(256.0)).astype(f32)) * f32(1.0 / 256.0)).astype(f32)
    return ft.astype(np.int64), w


def test_interpolated_tap_pair_is_a_fixed_texel_pair_with_a_weight():
    """blur_interp.h psx_lit_weight (GaussMode VLFeat_Relative, s_pyramid_build_ai.cu:17-69): the fetch at c - off / c + off with
    off = offset + (1 - u) in [offset, offset + 1] reads the FIXED texel pair (c - offset - 1, c - offset) / (c + offset,
    c + offset + 1); when the float arithmetic lands on the second texel (floor = kstat + 1) its weight is 0, which is the value
    of weight 1 on the fixed pair.  Every column up to 16384, every pair offset up to 15, u over its range incl. 0 and 1 and
    values that round c -+ off onto an integer."""
    rng = np.random.default_rng(11)
    c = np.arange(0, 16384, dtype=np.int64)
    cf = c.astype(f32)
    us = np.concatenate([np.array([0.0, 1.0, 0.5, 1e-7, 1.0 - 6e-8, 0.25, 0.75], dtype=f32), rng.random(40).astype(f32)])
    changes = 0
    for offset in range(1, 17, 2):
        for u in us:
            off = f32(f32(offset) + (f32(1.0) - u))               # offset + (1.0f - u): int -> float, one addition
            for right in (False, True):
                t = (cf + off).astype(f32) if right else (cf - off).astype(f32)
                ts = (t + f32(0.5)).astype(f32)                    # readTex adds 0.5 ...
                tb = (ts - f32(0.5)).astype(f32)                   # ... the unit takes it off again
                k, w = _tex_axis(tb)
                kstat = c + offset if right else c - offset - 1
                assert np.all((k == kstat) | (k == kstat + 1)), (offset, float(u), right)
                assert np.all(w[k == kstat + 1] == 0.0), (offset, float(u), right)
                assert np.all((w >= 0.0) & (w <= 1.0))
                wfix = np.where(k == kstat, w, f32(1.0))
                changes += int(np.count_nonzero(np.diff(wfix)))
    # the weight on the fixed pair does change with the coordinate (at binade boundaries of c -+ off): rarely, but it happens --
    # which is why the kernel surveys it per workgroup and keeps a literal per-element path
    assert 0 < changes < 4000, changes


def test_x2_upsampling_puts_the_fixed_span_fetches_on_the_half_texel_grid():
    """pyramid_fixed.hip, octave 0 of GaussMode Fixed9 / Fixed15 at W = 2w (s_pyramid_fixed.cu:123-140): the fetch of column cx
    at tap i is tex2D((cx + 1) * (1 / W), fma(i, 1 / H, (y + 1) * (1 / H))) of the normalised, clamped, linear-filtered image.
    In float these coordinates land on texel t = (n)/2 - ... with a 1.8 weight of exactly 0 or 1/2 -- or, on the knife edge of an
    integer texel coordinate, on (i0 - 1, weight 1), which is the same texel value -- for every width up to 4096 texels (the bound
    psx_fixed_octave0_ok enforces).  The kernel's constant-weight form rests on this."""
    for w in (4, 5, 7, 64, 333, 960, 1171, 1920, 2100, 4095, 4096):
        W = 2 * w
        mul = (f32(1.0) / f32(W)).astype(f32) if isinstance(f32(1.0) / f32(W), np.ndarray) else f32(f32(1.0) / f32(W))
        # columns: xpos = (cx + tshift) * mul_w with tshift = 1.0
        cx = np.arange(-8, W + 8, dtype=np.int64)
        xpos = ((cx.astype(f32) + f32(1.0)).astype(f32) * mul).astype(f32)
        tb = ((xpos * f32(w)).astype(f32) - f32(0.5)).astype(f32)
        i0, a = _tex_axis(tb)
        n = cx + 1                                                  # exact texel coordinate = n / 2 - 1/2
        even = (n % 2) == 0                                         # n even: half-texel position, weight 1/2 on (n/2 - 1, n/2)
        assert np.all(a[even] == 0.5) and np.all(i0[even] == n[even] // 2 - 1), w
        odd = ~even                                                 # n odd: integer texel (n - 1) / 2: weight 0 there, or weight 1 one below
        m = (n[odd] - 1) // 2
        ok = ((i0[odd] == m) & (a[odd] == 0.0)) | ((i0[odd] == m - 1) & (a[odd] == 1.0))
        assert np.all(ok), w
        # rows of tap k: vn = fma(k, mul_h, ypos) -- one rounding; same grid (H = 2h, here h = w)
        y = np.arange(0, W, 37, dtype=np.int64)
        ypos = ((y.astype(f32) + f32(1.0)).astype(f32) * mul).astype(f32)
        for k in range(-7, 8):
            vn = (np.full(y.shape, k, np.float64) * np.float64(mul) + ypos.astype(np.float64)).astype(f32)
            tbv = ((vn * f32(w)).astype(f32) - f32(0.5)).astype(f32)
            j0, b = _tex_axis(tbv)
            nn = y + k + 1
            ev = (nn % 2) == 0
            assert np.all(b[ev] == 0.5) and np.all(j0[ev] == nn[ev] // 2 - 1), (w, k)
            mm = (nn[~ev] - 1) // 2
            assert np.all(((j0[~ev] == mm) & (b[~ev] == 0.0)) | ((j0[~ev] == mm - 1) & (b[~ev] == 1.0))), (w, k)
