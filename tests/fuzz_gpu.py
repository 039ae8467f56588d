"""Randomised parity soak (not collected by pytest: run by hand on the GPU box):
    python tests/fuzz_gpu.py [seconds] [seed]
Random ragged batches of varied material and random descriptor masks through the C-ABI against the oracle,
with the tolerances of tests/_tol.py; prints every mismatch and exits non-zero if there was one."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afec_amd as afx                                   # noqa: E402
from tests import _tol                                   # noqa: E402
from tests._oracle import FIELDS, NEIGH_FIELDS, Oracle  # noqa: E402


def material(rng, n):
    t = np.arange(n) / 44100.0
    kind = rng.integers(0, 9)
    if kind == 0:
        x = rng.uniform(-1, 1, n)
    elif kind == 1:
        x = rng.uniform(0.05, 1.0) * np.sin(2 * np.pi * rng.uniform(30, 8000) * t + rng.uniform(0, 6.28))
    elif kind == 2:
        x = sum(rng.uniform(0.05, 0.4) * np.sin(2 * np.pi * rng.uniform(50, 5000) * t) for _ in range(rng.integers(2, 6)))
    elif kind == 3:
        x = rng.standard_normal(n) * np.maximum(np.exp(-t / rng.uniform(0.01, 0.5)), 1e-9)   # -180 dB floor
    elif kind == 4:
        x = np.zeros(n)
        for _ in range(rng.integers(1, 6)):
            a = int(rng.integers(0, n))
            m = int(min(n - a, rng.integers(100, 20000)))
            x[a:a + m] += rng.uniform(0.1, 0.8) * np.sin(2 * np.pi * rng.uniform(80, 2000) * np.arange(m) / 44100.0) * np.exp(-np.arange(m) / rng.uniform(500, 8000))
    elif kind == 5:
        x = np.clip(3.0 * np.sin(2 * np.pi * rng.uniform(60, 900) * t), -1, 1)      # clipped: plateaus in time
    elif kind == 6:
        x = 0.3 * rng.standard_normal(n) + rng.uniform(-0.5, 0.5)                   # DC offset
    elif kind == 7:
        x = 1e-4 * rng.standard_normal(n)                                            # near the silence threshold
        x *= 10 ** rng.uniform(-1, 1.5)
    else:
        x = np.sign(np.sin(2 * np.pi * rng.uniform(40, 1500) * t)) * rng.uniform(0.1, 1.0)
    if rng.random() < 0.3:
        a = int(rng.integers(0, n))
        x[a:a + int(rng.integers(0, 30000))] = 0.0                                  # a silent gap
    return x


def dump(round_index, bufs, mask):
    """failing inputs go to gpurun_out/fuzz/ for analysis against the oracle and the reference driver"""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fuzz")
    os.makedirs(d, exist_ok=True)
    np.savez_compressed(os.path.join(d, f"round{round_index}.npz"), mask=np.array(mask),
                        **{f"buf{i}": b for i, b in enumerate(bufs)})


def draw_round(rng, stats_mode=False):
    """Every random draw of one round, in the order the loop has always made them -- none depends on a GPU result, so a
    round's inputs can be regenerated without a GPU (tools/fuzz_replay.py fast-forwards a seed to a round with this)."""
    bufs = []
    for _ in range(int(rng.integers(1, 7))):
        n = int(rng.choice([rng.integers(0, 3000), rng.integers(2048, 40000), rng.integers(2048, 200000)]))
        x = material(rng, n) if n else np.zeros(0)
        bufs.append(x.astype(np.float32 if rng.random() < 0.5 else np.float64))
    dt = bufs[0].dtype
    bufs = [b.astype(dt) for b in bufs]
    mask = int(rng.integers(1, 1 << 22)) & afx.D_ALL_PER_FRAME
    if mask == 0:
        mask = afx.D_ALL_PER_FRAME
    if stats_mode:
        # the statistics class of the half-wave kernel (run with AFX_FUZZ_KERNEL=halfwave): MFCC + a random subset of
        # spectral rms / centroid / spread / skewness / kurtosis / rolloff / flatness, float32 PCM
        mask = afx.D_MFCC | (int(rng.integers(0, 128)) << 1)
        bufs = [b.astype(np.float32) for b in bufs]
    out = {"bufs": bufs, "mask": mask, "statistics": False, "load_files": None, "rhythm_info": None}
    if sum((b.size - 2048) // 1024 + 1 for b in bufs if b.size >= 2048) == 0:
        return out          # a round without a frame ends here
    out["statistics"] = bool(rng.random() < 0.25)
    if rng.random() < 0.2:
        files = []
        for _ in range(int(rng.integers(1, 5))):
            ch = int(rng.integers(1, 9))
            nfr = int(rng.choice([rng.integers(1, 3000), rng.integers(2048, 60000)]))
            y = np.stack([material(rng, nfr) * rng.uniform(0.05, 1.2) for _ in range(ch)], axis=1)
            files.append((y, ch, int(rng.integers(0, 3))))
        out["load_files"] = files
    if rng.random() < 0.5 and any(b.size >= 512 for b in bufs):
        out["rhythm_info"] = [(44100, int(rng.integers(-3000, 1)), int(b.size * rng.choice([1, 1, 2]))) for b in bufs]
    return out


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    # AFX_FUZZ_KERNEL = wave64 | halfwave forces the STFT kernel's layout (afx_plan_desc.frame_kernel); default: by batch size
    kernel = {"wave64": afx.FRAME_KERNEL_WAVE64, "halfwave": afx.FRAME_KERNEL_HALFWAVE}.get(os.environ.get("AFX_FUZZ_KERNEL", ""), afx.FRAME_KERNEL_AUTO)
    plan, oracle = afx.Plan(max_analysis_ms=0, frame_kernel=kernel), Oracle()
    t0 = time.time()
    rounds = frames = bad = skipped = rhythm_frames = rhythm_gate_flips = 0
    while time.time() - t0 < seconds:
        drawn = draw_round(rng, bool(os.environ.get("AFX_FUZZ_STATS")))
        bufs, mask = drawn["bufs"], drawn["mask"]
        res = plan.extract(bufs, mask)
        ref = np.concatenate([oracle.run(b.astype(np.float64)) for b in bufs]) if bufs else None
        nref = np.concatenate([oracle.run_neighbours(b.astype(np.float64)) for b in bufs])
        frames += ref.shape[0]
        if ref.shape[0] == 0:
            rounds += 1
            continue
        # Frames whose discrete outputs are decided by rounding noise in ANY implementation (the reference's own
        # objects disagree with the oracle there): an exactly flat magnitude spectrum (an impulse at the frame
        # start: "strict local maximum" is a coin toss per bin) and a flat difference function (a frame that is
        # almost entirely digital silence: yinfast's confidence is ~1e-14 and its arg-min picks noise).
        mags = ref[:, 1:752]
        flat_spectrum = (mags.max(axis=1) - mags.min(axis=1)) <= 1e-9 * mags.max(axis=1)
        # ... and, for the count of strict local maxima per sub-band only, a frame in which two neighbouring bins differ by
        # less than the FFT's own rounding (32 eps of the frame's largest bin): the far skirt of a loud tone is smooth at
        # 1e-10 of the peak and where it turns, neighbours differ by 1e-16..1e-18 of it (seed 99, round 12373: band 13 of a
        # frame with peak 0.175 has bins 546 / 547 at 5.06005024e-11 / 5.06005006e-11; the oracle counts 6 maxima, the
        # half-wave kernel 7).  0.4 % of the frames.
        near_tie = np.abs(np.diff(ref[:, 0:753], axis=1)).min(axis=1) <= 32 * 2.2e-16 * ref[:, 0:753].max(axis=1)
        # ... or whose first half is 120 dB below the frame: yinfast takes d(tau) = E1 + E2(tau) - 2 C(tau) with the
        # correlation C from an FFT of the whole frame, whose rounding is ~1e-16 of the FRAME's energy; a burst that starts
        # in the second half behind a tail at 1e-15 (seed 95, round 18824: first half rms 1.2e-15, second 0.15) leaves
        # d(tau) = 1e-28 for small tau under 1e-15 of noise -- the direct sum gives 1901.6 Hz, the oracle 1922.3, the GPU
        # 1922.0, each at "confidence 1".
        zero_counts, quiet_first_half = [], []
        for b in bufs:
            nf = oracle.num_frames(b.size, False)
            cz = np.concatenate([[0], np.cumsum(b == 0)])
            hop_energy = (b[:(b.size // 1024) * 1024].astype(np.float64).reshape(-1, 1024) ** 2).sum(axis=1)   # frame f = hops f, f + 1
            zero_counts += [cz[1024 * f + 2048] - cz[1024 * f] for f in range(nf)]
            quiet_first_half += [hop_energy[f] <= 1e-12 * (hop_energy[f] + hop_energy[f + 1]) for f in range(nf)]
        flat_yin = (np.array(zero_counts, dtype=np.int64) >= 1024) | np.array(quiet_first_half, dtype=bool)   # at least half of the frame is digital silence
        skipped += int(flat_spectrum.sum() + flat_yin.sum() + (near_tie & ~flat_spectrum).sum())
        discrete_spectral = {"sub_complexity", "sub_flux", "spectral_flux", "spectral_complexity"}
        pitch_fields = {"f0", "failsafe_f0"}
        for field, (a, b) in FIELDS.items():
            if field == "mag" or field not in res:
                continue
            keep = ~flat_spectrum if field in discrete_spectral else np.ones(ref.shape[0], bool)
            if field == "sub_complexity":
                keep = keep & ~near_tie
            try:
                _tol.check_gpu(field, res[field].reshape(ref.shape[0], -1)[keep], ref[keep, a:b], *_tol.GPU_TOL[field], what=f"seed {seed} round {rounds}: ", ceiling_is_fatal=False)
            except AssertionError as e:
                bad += 1
                dump(rounds, bufs, mask)
                print(f"round {rounds} mask {mask:#x}: {e}")
        for field, col in NEIGH_FIELDS.items():
            if field not in res:
                continue
            keep = ~flat_spectrum if field in discrete_spectral else (~flat_yin if field in pitch_fields else np.ones(ref.shape[0], bool))
            try:
                _tol.check_gpu(field, res[field][keep], nref[keep, col], *_tol.NEIGH_TOL[field], what=f"seed {seed} round {rounds}: ", ceiling_is_fatal=False)
            except AssertionError as e:
                bad += 1
                dump(rounds, bufs, mask)
                print(f"round {rounds} mask {mask:#x}: {e}")
        # per-file statistics of the GPU's own series (the reduction alone is compared, at 1e-9)
        if drawn["statistics"]:
            from tests import _oracle
            b = plan.batch(bufs, mask | afx.D_STATISTICS)
            b.run()
            series, st = b.fetch(), b.fetch_statistics()
            b.close()
            off = series["frame_offset"]
            for i in range(len(bufs)):
                if off[i + 1] - off[i] < 2 or off[i + 1] - off[i] > 1024:
                    continue
                for name, arr in st.items():
                    if name == "stats_status":
                        continue
                    vals = series[name][off[i]:off[i + 1]].reshape(off[i + 1] - off[i], -1)
                    got = arr[i].reshape(vals.shape[1], 13)
                    for w in range(vals.shape[1]):
                        v = vals[:, w]
                        if not np.all(np.isfinite(v)):
                            continue
                        want = _oracle.calc_statistics(v)
                        tiny = abs(v.sum()) < 1e-3 * np.abs(v).sum() or not np.abs(v).sum() > 0   # sum by cancellation: centroid, spread, ... are ill-conditioned
                        # a series that is constant to 1e-9 of its level: the rounding of the mean is a visible part of
                        # the standard deviation, which skewness and kurtosis divide by to the 3rd / 4th power (observed:
                        # 4.60e35 vs 4.57e35)
                        flatline = float(np.std(v)) <= 1e-9 * float(np.abs(v).max())
                        # TStatistics::Centroid / Spread (Statistics.cpp:459-506) are quotients of index-weighted sums of the
                        # VALUES: for a series of mixed sign (band flux values of +-1) those sums can cancel to rounding
                        # residue although the plain sum does not (observed: spread 2.4e-12 from terms of size 1e3, which a
                        # 1e-15 perturbation of the series moves by 5 %, skewness = mean(((x - c) / spread)^3) by 15 %)
                        jdx = np.arange(v.size, dtype=np.float64)
                        with np.errstate(all="ignore"):
                            cen = (jdx * v).sum() / v.sum() if v.sum() != 0 else 0.0
                            terms = (jdx - cen) ** 2 * v
                            cancelling = (np.abs(terms).sum() > 1e6 * abs(terms.sum())) or (np.abs(jdx * v).sum() > 1e6 * abs((jdx * v).sum()))
                        for j, sn in enumerate(afx.STAT_NAMES):
                            if tiny and sn in ("centroid", "spread", "skewness", "kurtosis", "flatness"):
                                continue
                            if flatline and sn in ("skewness", "kurtosis"):
                                continue
                            if cancelling and sn in ("centroid", "spread", "skewness", "kurtosis"):
                                continue
                            if not abs(got[w, j] - want[j]) <= 1e-8 * abs(want[j]) + 1e-11:
                                bad += 1
                                dump(rounds, bufs, mask)
                                print(f"round {rounds} statistics {name}[{w}].{sn} of buffer {i}: got {got[w, j]!r} want {want[j]!r}")
        # the LoadSample front end: random decoded files, samples / offsets / peak bit-exact against the oracle
        if drawn["load_files"] is not None:
            from tests import _oracle
            files = []
            for y, ch, fmt in drawn["load_files"]:
                if fmt == 0:
                    data = np.clip(np.round(y * 32767), -32768, 32767).astype(np.int16).reshape(-1)
                elif fmt == 1:
                    v = np.clip(np.round(y * 8388607), -8388608, 8388607).astype(np.int32).reshape(-1)
                    data = np.zeros((v.size, 3), dtype=np.uint8)
                    data[:, 0] = v & 0xFF; data[:, 1] = (v >> 8) & 0xFF; data[:, 2] = (v >> 16) & 0xFF
                    data = data.reshape(-1)
                else:
                    data = y.astype(np.float32).reshape(-1)
                files.append((data, ch))
            b, infos = plan.batch_from_raw(files, afx.D_MFCC)
            for i, (data, ch) in enumerate(files):
                want, winfo = _oracle.load_sample(data, ch)
                ok = all(infos[i][k] == winfo[k] for k in ("data_offset", "silent_leading", "silent_trailing", "n_samples", "peak_value"))
                nf = plan.num_frames(len(want))
                kept = (nf - 1) * 1024 + 2048 if nf > 0 else 0
                ok = ok and np.array_equal(b.fetch_samples(i, kept), want[:kept])
                ok = ok and abs(infos[i]["rms_value"] - winfo["rms_value"]) <= 3e-7 * winfo["rms_value"] + 1e-12
                if not ok:
                    bad += 1
                    print(f"round {rounds} load front end: file {i} ({data.dtype}, {ch} ch): {infos[i]} vs {winfo}")
            b.close()
        # the rhythm tracker on the round's buffers (float64 here: the oracle sees the very same samples): onset frames
        # identical, onset functions to a few float ulps of the file's peak, scalars to 1e-5; a detection that differs
        # is reported with its margin to the threshold
        if drawn["rhythm_info"] is not None:
            rb = plan.batch(bufs, afx.D_RHYTHM)
            info = drawn["rhythm_info"]
            rb.set_file_info(info)
            rb.run()
            rr = rb.fetch_rhythm(onset_functions=True)
            rb.close()
            off = rr["offsets"]
            for i, b in enumerate(bufs):
                want = oracle.run_rhythm(b.astype(np.float64), original_samples=info[i][2], data_offset=info[i][1])
                sl = slice(off[i], off[i + 1])
                rhythm_frames += int(off[i + 1] - off[i])
                if off[i + 1] == off[i]:
                    continue
                odf, wodf = rr["onset_functions"][sl].astype(np.float64), want["odf"].T.astype(np.float32).astype(np.float64)
                scale = np.abs(wodf).max(axis=0) + 1e-30
                # The rectified complex-domain function sums a bin's deviation only when its whitened magnitude did not
                # fall (OnsetDetector.cpp:419): a bin that sits exactly at its follower's maximum has the float value
                # 1.0 in one FFT implementation and 1 - 2^-24 in another (the reference's IPP / vDSP / Ooura builds
                # differ the same way), so a single term may enter or leave a frame's sum.  Such frames (at most 0.2 %)
                # may deviate by a few terms' worth; everything else to a few float ulps.
                dev_frames = np.abs(odf - wodf) > 2e-6 * np.abs(wodf) + 2e-7 * scale
                flips = int(dev_frames[:, 0].sum())
                ok = (not dev_frames[:, 1].any() and flips <= max(1, int(0.002 * odf.shape[0])) and
                      np.all(np.abs(odf[:, 0] - wodf[:, 0]) <= 0.1))
                rhythm_gate_flips += flips
                for t in range(2):
                    ok = ok and np.array_equal(np.nonzero(rr["onsets"][sl, t])[0], np.nonzero(want["onsets"][t])[0])
                finite = np.isfinite(want["scalars"])
                ok = ok and np.array_equal(np.isfinite(rr["scalars"][i]), finite)
                ok = ok and np.all(np.abs(rr["scalars"][i][finite] - want["scalars"][finite]) <= 1e-5 * np.abs(want["scalars"][finite]) + 1e-9)
                if not ok:
                    bad += 1
                    dump(rounds, bufs, afx.D_RHYTHM)
                    print(f"round {rounds} rhythm tracker, buffer {i} ({b.size} samples): scalars {rr['scalars'][i]} vs {want['scalars']}; "
                          f"max onset-function error {np.max(np.abs(odf - wodf) / scale):.2e}")
        rounds += 1
    print(f"{rounds} rounds, {frames} frames, {rhythm_frames} rhythm frames ({rhythm_gate_flips} with a rectification-gate flip), "
          f"{bad} mismatching (round, descriptor) pairs, "
          f"{skipped} ill-conditioned frames left out of the discrete comparisons, seed {seed}")
    if _tol.ceiling_warnings:
        print(f"{len(_tol.ceiling_warnings)} value(s) inside the bar but above a regression ceiling (warnings, not mismatches): "
              + "; ".join(f"{w[0]}{w[1]} {w[2]:.2e} > {w[3]:g}" for w in _tol.ceiling_warnings[:8]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
