"""CPU checks of arithmetic identities the HIP kernels rely on (numpy emulation of the fp32 instruction sequences)."""
import numpy as np


def _fma32(a, b, c):
    """fmaf for float32 arrays: the product of two float32 is exact in float64, one rounding to float32 at the end
    (the float64 addition is exact or far below a float32 half-ulp for the magnitudes used here)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_uniform_division_by_reciprocal_is_exact():
    """psv_variance_reuse_kernel divides by the wave-uniform W - 1 / H - 1 as q0 = x * rc, q1 = fma(fma(-q0, c, x), rc, q0)
    with rc = RN(1 / c).  Markstein: q1 is the correctly rounded x / c unless c's significand is all ones.  Checked against
    IEEE float32 division for image extents from 4 to 4096 and operands from 1e-3 to 1e7 (coordinates of points far
    behind a camera included)."""
    rng = np.random.default_rng(0)
    for c in (3, 7, 63, 79, 255, 319, 479, 639, 959, 1079, 1279, 1919, 4095):
        c32 = np.float32(c)
        rc = np.float32(1.0 / np.float64(c))
        x = (rng.standard_normal(400_000) * rng.choice([1e-3, 1.0, 50.0, 400.0, 1e4, 1e7], 400_000)).astype(np.float32)
        q0 = (x * rc).astype(np.float32)
        q1 = _fma32(_fma32(-q0, np.full_like(x, c32), x), np.full_like(x, rc), q0)
        assert np.array_equal(q1, (x / c32).astype(np.float32)), c


def test_power_of_two_mean_is_a_multiplication():
    """torch_scatter mean = sum / count; for a power-of-two count the kernels multiply by 1 / count instead."""
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200_000) * rng.choice([1e-6, 1e-2, 1.0, 30.0], 200_000)).astype(np.float32)
    for n in (1, 2, 4, 8, 16):
        assert np.array_equal((x / np.float32(n)).astype(np.float32), (x * np.float32(1.0 / n)).astype(np.float32))


def test_bf16_split_carries_sixteen_mantissa_bits():
    """x = hi + lo with hi = RNE_bf16(x), lo = RNE_bf16(x - hi): |x - (hi + lo)| <= 2^-17 |x| (the operand precision of the
    split-bf16 MFMA kernels), and hi + lo is exactly representable in float32."""
    def rne_bf16(v):
        u = v.view(np.uint32).astype(np.uint64)
        return ((((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16) & 0xffffffff).astype(np.uint32).view(np.float32)
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(300_000) * rng.choice([1e-4, 1.0, 100.0], 300_000)).astype(np.float32)
    hi = rne_bf16(x)
    lo = rne_bf16((x - hi).astype(np.float32))
    rec = (hi.astype(np.float64) + lo.astype(np.float64))
    assert np.array_equal(rec.astype(np.float32).astype(np.float64), rec)
    assert np.all(np.abs(rec - x.astype(np.float64)) <= 2.0 ** -17 * np.abs(x.astype(np.float64)) + 1e-45)
