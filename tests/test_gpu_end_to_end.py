"""End to end at BASELINE.json's C3 / C4 shapes: decoded PCM -> LoadSample front end (f3) -> the
spectral path -> per-file statistics (f1), all on the GPU through the C-ABI, against the oracle
pipeline (load_sample -> run -> calc_statistics) on a sample of the files."""
import numpy as np
import pytest

import afec_amd as afx
from tests import _oracle, _tol
from tests._oracle import FIELDS, Oracle

pytestmark = pytest.mark.gpu


def synth_file(rng, seconds, stereo):
    n = int(44100 * seconds)
    t = np.arange(n) / 44100.0
    x = np.zeros(n)
    for _ in range(int(rng.integers(1, 4))):
        x += rng.uniform(0.2, 0.6) * np.sin(2 * np.pi * rng.uniform(110.0, 4000.0) * t + rng.uniform(0, 6.28))
    x += rng.uniform(0.2, 0.8) * rng.uniform(-1, 1, n) * np.exp(-t / rng.uniform(0.05, 0.5))
    x[:2205] = 0.0                                   # 50 ms of leading silence
    x *= rng.uniform(0.1, 0.9) / np.max(np.abs(x))
    if stereo:                                       # R = L delayed 7 samples x 0.8
        y = np.stack([x, 0.8 * np.roll(x, 7)], axis=1)
        return (y * 32767).astype(np.int16), 2
    return (x * 32767).astype(np.int16), 1


def check_properties_of_every_file(res, stats, infos, n_files, ora):
    """Size-independent properties, checked on ALL files of the batch (the oracle comparison below samples)."""
    off = res["frame_offset"]
    counts = np.diff(off)
    for i in range(n_files):
        assert counts[i] == ora.num_frames(infos[i]["n_samples"], cap=True), i          # the frame-count rule, SA:760-764, 814
    total = int(off[-1])
    for field in FIELDS:
        if field != "mag":
            assert np.all(np.isfinite(res[field])), field
    assert np.all(res["spectral_rms"] >= 0) and np.all(res["amplitude_rms"] >= 0)
    assert np.all(res["amplitude_peak"] <= 1.0) and np.all(res["amplitude_rms"] <= res["amplitude_peak"] + 1e-15)
    roll = res["spectral_rolloff"]
    assert np.all(roll == np.round(roll / 43.0) * 43.0) and np.all((roll >= 0) & (roll <= 738 * 43))   # 43 x a bin count
    cplx = res["sub_complexity"]
    assert np.all(cplx == np.round(cplx)) and np.all(cplx >= 0)
    assert np.all(np.abs(res["spectral_flux"]) <= 1.0 + 1e-12) and np.all(np.abs(res["sub_flux"]) <= 1.0 + 1e-12)   # Pearson r
    assert np.all((res["spectral_flatness"] >= 0) & (res["spectral_flatness"] <= 1.0))
    assert np.all((res["spectral_centroid"] >= 0) & (res["spectral_centroid"] <= 737.0))
    # a frame's 28 band energies are sums of |X|^2 over disjoint bin ranges of bins 1..1023
    assert np.all(res["spectrum_bands"] >= 0)
    # first frame of every file: flux against itself (SA:937-940) is 1 unless the frame is silent
    first = res["spectral_flux"].reshape(total)[off[:-1]]
    assert np.all((np.abs(first - 1.0) < 1e-9) | (first == 0.0))
    # statistics: min <= median, mean <= max; variance >= 0; series of one frame keep the initial zeros (Statistics.cpp:72-89)
    for field, (a, b) in FIELDS.items():
        if field == "mag":
            continue
        st = stats[field].reshape(n_files, b - a, 13)
        assert np.all(np.isfinite(st)), field
        mn, mx, med, mean, var = st[..., 0], st[..., 1], st[..., 2], st[..., 3], st[..., 5]
        span = 1e-9 * (1.0 + np.abs(mx))
        assert np.all(mn <= med + span) and np.all(med <= mx + span) and np.all(mn <= mean + span) and np.all(mean <= mx + span), field
        assert np.all(var >= 0), field


@pytest.mark.parametrize("shape,n_files,n_sampled", [("c3_mono_2s", 1000, 32), ("c4_stereo_1s", 300, 8)])
def test_pipeline_matches_oracle_on_sampled_files(shape, n_files, n_sampled):
    """c3: BASELINE.json configs[2] at its full size -- 1000 synthetic 2.0 s files in ONE batch; c4: a 300-file slice of
    configs[3]'s per-GPU share (the share itself, 12 500 files, runs in bench.py).  Properties on every file, the oracle
    pipeline on a sample."""
    rng = np.random.default_rng(51)
    stereo = shape.startswith("c4")
    files = [synth_file(rng, 1.0 if stereo else 2.0, stereo) for _ in range(n_files)]
    plan = afx.Plan()
    mask = afx.D_ALL_LOW_LEVEL | afx.D_STATISTICS
    batch, infos = plan.batch_from_raw(files, mask)
    batch.run()
    res = batch.fetch()
    stats = batch.fetch_statistics()
    assert stats["stats_status"].tolist() == [0] * n_files and res["buf_status"].tolist() == [0] * n_files
    off = res["frame_offset"]
    ora = Oracle()
    check_properties_of_every_file(res, stats, infos, n_files, ora)
    for i in rng.choice(n_files, n_sampled, replace=False):
        data, ch = files[i]
        mono, info = _oracle.load_sample(data, ch)
        assert infos[i]["n_samples"] == info["n_samples"] and infos[i]["data_offset"] == info["data_offset"]
        ref = ora.run(mono, cap=True)
        assert off[i + 1] - off[i] == ref.shape[0]
        for field, (a, b) in FIELDS.items():
            if field == "mag":
                continue
            rtol, atol = _tol.GPU_TOL[field]
            got = res[field][off[i]:off[i + 1]].reshape(ref.shape[0], -1)
            _tol.check_gpu(field, got, ref[:, a:b], rtol, atol, what=f"{shape} file {i} ")
            # per-file statistics of the same series
            width = b - a
            gs = stats[field].reshape(n_files, width, 13)[i]
            for w in range(width):
                want = _oracle.calc_statistics(ref[:, a + w], np.zeros(13))
                # observed (profiles/r02/parity_report.md): <= 6e-10 relative on every statistic, the series themselves
                # agree to ~2e-7 at worst (MFCC); 20 x the observed worst case
                tol = np.full(13, 1e-8)
                scale = 1e-6 * (1.0 + np.max(np.abs(want)))
                if field in ("spectral_rolloff", "sub_complexity"):
                    continue                          # discrete series: compared above frame by frame
                err = np.abs(gs[w] - want)
                ok = err <= tol * np.abs(want) + scale
                # TStatistics::Centroid and Spread divide by the *sum* of the series (Statistics.cpp:459-506): for a series
                # that sums to rounding residue (e.g. band flux values of +-1) they are decided by the last bit of the
                # inputs in any implementation -- and so are skewness / kurtosis, which are functions of exactly those
                # two (Statistics.cpp:510-554: ((x - centroid) / spread)^3, ^4).  Nothing else is waived.
                series = ref[:, a + w]
                if abs(series.sum()) < 1e-6 * np.abs(series).sum():
                    ok[[6, 7, 8, 9]] = True
                assert np.all(ok), (shape, int(i), field, w, gs[w], want)
    batch.close()
    plan.close()



def test_whole_file_whitening_chunks_match_the_oracle():
    """Batches of >= 768 short files: the whitening kernels walk every file as one chunk from the follower's reset state
    and follow_kernel is not launched (afx_batch_plan.cpp: cut_whitening_chunks) -- the path bench.py's C3 / C4 chains
    take.  900 one-second files with every per-frame descriptor; the whitened-spectrum neighbours (spectral complexity,
    fail-safe f0, ...) and the rest of a sample of the files against the oracle."""
    from tests._oracle import NEIGH_FIELDS
    rng = np.random.default_rng(52)
    files = [synth_file(rng, 1.0, False) for _ in range(900)]
    plan = afx.Plan()
    batch, infos = plan.batch_from_raw(files, afx.D_ALL_PER_FRAME)
    batch.run()
    res = batch.fetch()
    off = res["frame_offset"]
    ora = Oracle()
    for i in rng.choice(900, 12, replace=False):
        mono, _ = _oracle.load_sample(*files[i])
        ref, nref = ora.run(mono, cap=True), ora.run_neighbours(mono, cap=True)
        assert off[i + 1] - off[i] == ref.shape[0] <= 128
        for field, col in NEIGH_FIELDS.items():
            _tol.check_gpu(field, res[field][off[i]:off[i + 1]], nref[:, col], what=f"file {i} ")
        for field, (a, b) in FIELDS.items():
            if field != "mag":
                _tol.check_gpu(field, res[field][off[i]:off[i + 1]].reshape(ref.shape[0], -1), ref[:, a:b], what=f"file {i} ")
    batch.close()
    plan.close()
