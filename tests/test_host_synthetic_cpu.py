"""The host library's synthetic-input and sharding helpers (no GPU): BASELINE configs[1]'s generator and the crawler's
choice of worker threads per device."""
import numpy as np

from afec_amd import hostlib


def _restated(n, seed):
    """std::uniform_real_distribution<float>(-1, 1) over std::mt19937(seed) as libstdc++ evaluates it: one 32-bit word per
    float, float(word) / 2^32 (1.0 -> the float below it), x 2 - 1 in float; numpy's MT19937 under init_genrand seeding"""
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)
    u = bg.random_raw(n).astype(np.uint32).astype(np.float32) / np.float32(4294967296.0)
    u = np.where(u >= np.float32(1.0), np.nextafter(np.float32(1.0), np.float32(0.0)), u).astype(np.float32)
    return (np.float32(2.0) * u + np.float32(-1.0)).astype(np.float32)


def test_the_c2_generator_is_std_mt19937_1234():
    """x = U(-1, 1) from std::mt19937(1234).  Known answers: the 10 000th output of mt19937() seeded 5489 is
    4123659995 (the C++ standard's own check value); seed 1234's first floats as the C++ library produces them."""
    bg = np.random.MT19937()
    bg._legacy_seeding(5489)
    assert int(bg.random_raw(10000)[-1]) == 4123659995          # [rand.predef]: the restatement's twister is std::mt19937
    x = hostlib.fill_uniform_mt19937(1 << 16, 1234)
    assert x.dtype == np.float32 and x.shape == (1 << 16,)
    np.testing.assert_array_equal(x, _restated(1 << 16, 1234))
    np.testing.assert_allclose(x[:4], [-0.6169611, -0.00467265, 0.24421751, 0.63567686], rtol=0, atol=1e-7)
    assert -1.0 <= x.min() and x.max() < 1.0 and abs(float(x.mean())) < 0.01
    # another seed is another stream; bench.py's numpy fallback is the same arithmetic
    y = hostlib.fill_uniform_mt19937(4096, 1235)
    assert not np.array_equal(y, x[:4096])
    np.testing.assert_array_equal(y, _restated(4096, 1235))


def test_workers_per_device_follow_the_usable_cpus():
    """TCrawlOptions::mWorkersPerDevice = 0: floor(usable CPUs / devices), at most 5, at least 1"""
    cpus = hostlib.usable_host_cpus()
    assert cpus >= 1
    for g in (1, 2, 4, 8, 64):
        assert hostlib.workers_per_device_for(g) == max(1, min(5, int(cpus // g)))
