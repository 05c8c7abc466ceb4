"""Pins the CPU oracle's math restatement (oracle/orc_math.h, Dist1D in oracle/pim_oracle.c)
against golden vectors produced by the REFERENCE's own headers.

tests/golden/ref_math.bin is written by oracle/ref_golden.c, which #includes
/root/reference/src/math/*.h, common/random.h, rendering/sampler.h, rendering/cubemap.h and
links math/dist1d.c, math/sphgauss.c, rendering/cubemap.c where they lie (oracle/Makefile
target `golden`, container only).  Both sides are built -O2 -ffp-contract=off without
-ffast-math, so the bar is BIT-EXACT on every word (floats compared by bit pattern)."""
import ctypes
import os
import struct

import numpy as np
import pytest

from tests.util import load_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_math.bin")


def read_records(path):
    with open(path, "rb") as f:
        data = f.read()
    magic, version, nrec = struct.unpack_from("<III", data, 0)
    assert magic == 0x474D4950 and version == 1
    off = 12
    recs = []
    for _ in range(nrec):
        (name_len,) = struct.unpack_from("<I", data, off)
        off += 4
        name = data[off:off + name_len].decode()
        off += (name_len + 3) & ~3
        ncases, nin, nout = struct.unpack_from("<III", data, off)
        off += 12
        words = np.frombuffer(data, dtype=np.uint32, count=ncases * (nin + nout), offset=off).reshape(ncases, nin + nout)
        off += 4 * ncases * (nin + nout)
        recs.append((name, nin, nout, words.copy()))
    assert off == len(data)
    return recs


RECORDS = read_records(GOLDEN)


def test_golden_file_covers_the_survey_list():
    names = {r[0] for r in RECORDS}
    # the pure functions of the path that must be pinned
    for need in ["pcg1", "pcg2", "pcg3", "pcg4", "prng_seq", "NormalToTBN", "MapSquareToDisk",
                 "SampleCosineHemisphere", "SampleGGXMicrofacet", "SampleBaryCoord", "SampleUnitSphere",
                 "SampleUnitHemisphere", "SampleNGon", "SampleGaussPixelFilter", "SampleFreePath",
                 "PowerHeuristic", "LightPdf", "GGXPdf", "F_Dielectric", "D_GTR", "V_SmithCorrelated",
                 "Fd_Burley", "F_0_F_90", "AlbedoToTransmittance", "reflect3", "refract3",
                 "GammaDecode_rgba8", "GammaEncode_rgba8", "UnpackEmission", "Bilinear", "Xy16ToNormalTs",
                 "Cubemap_CalcUv", "Grid", "Dist1D_Bake_Sample", "Dist1D_Update", "SG_Accumulate6",
                 "GradientNoise3", "FbmGradientNoise3", "MiePhase", "camera_basis", "trs_flatten",
                 "f2_tent", "Cubemap_CalcDir", "AmbCube_Fit", "brdf_lut_sample", "GGXEnergyCompensation", "lm_embed_fragment", "lm_chartnode",
                 "Cubemap_MipChain", "Cubemap_ReadColor_4", "Cubemap_BlendMip"]:
        assert need in names, need


@pytest.mark.parametrize("rec", RECORDS, ids=[r[0] for r in RECORDS])
def test_oracle_matches_reference_bits(rec):
    name, nin, nout, words = rec
    lib = load_oracle()
    fn = getattr(lib, "orc_hook_" + name)
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    out = np.zeros(nout, dtype=np.uint32)
    bad = []
    for i in range(words.shape[0]):
        inp = np.ascontiguousarray(words[i, :nin])
        fn(inp.ctypes.data, out.ctypes.data)
        exp = words[i, nin:]
        if not np.array_equal(out, exp):
            # +0 / -0 are the only tolerated difference in bit pattern (none observed)
            bad.append((i, inp.view(np.float32).tolist(), out.view(np.float32).tolist(), exp.view(np.float32).tolist()))
    assert not bad, f"{name}: {len(bad)} of {words.shape[0]} cases differ, first: {bad[0]}"


def _ulp_diff(a_bits, b_bits):
    """distance in units in the last place between two float32 bit patterns (monotone integer map)"""
    a = a_bits.astype(np.int64)
    b = b_bits.astype(np.int64)
    a = np.where(a & 0x80000000, 0x80000000 - a, a)
    b = np.where(b & 0x80000000, 0x80000000 - b, b)
    return np.abs(a - b)


# records whose outputs are all integers (must stay bit-exact in the portable variant too)
_INTEGER_RECORDS = {"pcg1", "pcg2", "pcg3", "pcg4", "GammaEncode_rgba8"}


@pytest.mark.parametrize("rec", RECORDS, ids=[r[0] for r in RECORDS])
def test_portable_oracle_matches_reference_within_ulps(rec):
    """liborc_p.so (portable sin/cos/log/log2/exp, oracle/orc_libm.h) against the same vectors.
    Bar: every float word within 4 ulp of the reference or within 2e-6 absolute (sin/cos near a
    zero crossing lose relative accuracy in any libm); integer words exact.  Records that use no
    transcendental function come out bit-exact and are checked as such by the ulp bound."""
    name, nin, nout, words = rec
    lib = load_oracle(portable=True)
    fn = getattr(lib, "orc_hook_" + name)
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    out = np.zeros(nout, dtype=np.uint32)
    worst = 0
    for i in range(words.shape[0]):
        inp = np.ascontiguousarray(words[i, :nin])
        fn(inp.ctypes.data, out.ctypes.data)
        exp = words[i, nin:]
        if np.array_equal(out, exp):
            continue
        assert name not in _INTEGER_RECORDS, f"{name}: integer record differs at case {i}"
        of, ef = out.view(np.float32).astype(np.float64), exp.view(np.float32).astype(np.float64)
        ulps = _ulp_diff(out, exp)
        ok = (ulps <= 4) | (np.abs(of - ef) <= 2e-6) | (np.isnan(of) & np.isnan(ef))
        assert ok.all(), f"{name} case {i}: got {of.tolist()} expected {ef.tolist()} (ulps {ulps.tolist()})"
        worst = max(worst, int(ulps[ok & (np.abs(of - ef) > 2e-6)].max(initial=0)))
    assert worst <= 4
