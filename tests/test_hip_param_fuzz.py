"""Randomized parameter sets against the reference.  The bare division / square-root sequences of
rs_math.hpp and the uniform reciprocals of rs_consts_dev.h are exact on a DOMAIN that the model's
parameters help to set (rs_consts_domain_error guards it at plan creation); the fixed cases of the
other tests sit at the reference's defaults.  Here every physical parameter the path reads moves
within a plausible band, together with the time step, the layer count, the output depth, the
initialization length and relaxation - and the outputs must still be the reference's, bit for bit
(oracle/_ref, built from the reference's own sources; the C restatement when that is absent)."""
import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi, lib

pytestmark = pytest.mark.gpu


def _kind():
    return "ref" if oh.have_ref() else "port"


# (name, low, high): multiplicative band around the default unless noted
_SCALED = [("CalmLimDay", 0.3, 2.0), ("CalmLimNgt", 0.3, 3.0), ("TrfFricNgt", 0.0, 2.0),
           ("TrFfricDay", 0.0, 2.0), ("LVap", 0.8, 1.2), ("LFus", 0.8, 1.2), ("WatDens", 0.95, 1.05),
           ("WatMHeat", 0.8, 1.2), ("PorEvaF", 0.5, 1.0), ("ZMom", 0.05, 2.0), ("ZHeat", 0.1, 10.0),
           ("Emiss", 0.9, 1.05), ("Albedo", 0.5, 3.0), ("MaxPormms", 0.5, 2.0), ("DampDpth", 0.7, 1.5),
           ("AZ", 0.5, 1.5), ("DampWearF", 0.2, 1.8), ("AlbSnow", 0.8, 1.3), ("vsh1", 0.7, 1.4),
           ("vsh2", 0.7, 1.4), ("Poro1", 0.5, 2.0), ("Poro2", 0.5, 1.5), ("RhoB1", 0.8, 1.2),
           ("RhoB2", 0.8, 1.2), ("Silt1", 0.5, 3.0), ("Silt2", 0.5, 1.2), ("WetSnowFormR", 0.5, 2.0),
           ("WetSnowMeltR", 0.5, 1.5), ("MaxSnowmms", 0.1, 2.0), ("MaxDepmms", 0.5, 2.0),
           ("MaxIcemms", 0.1, 2.0), ("MaxExtmms", 0.5, 2.0), ("Snow2IceFac", 0.5, 1.5)]
_SHIFTED = [("TClimG", -6.0, 6.0), ("freezing_limit_normal", -0.5, 0.4), ("snow_melting_limit_normal", -0.2, 1.0),
            ("ice_melting_limit_normal", -0.2, 1.0), ("frost_melting_limit_normal", -1.0, 1.0),
            ("frost_formation_limit_normal", -0.2, 1.0), ("T4Melt_normal", -0.2, 0.5),
            ("TLimColdH", -3.0, 3.0), ("TLimColdL", -3.0, 3.0), ("NightOn", -3.0, 3.0), ("NightOff", -2.0, 4.0)]


def _case(seed):
    rs = np.random.RandomState(1000 + seed)
    dt = float(rs.choice([10.0, 20.0, 30.0, 30.0, 60.0, 90.0]))
    spk = int(round(3600.0 / dt))
    L = int(rs.choice([6, 9, 12])) * spk + 1
    n = int(rs.choice([130, 192, 257]))
    f = oh.synth_forcing(n, L, seed=seed, steps_per_knot=spk)
    s = abi.default_settings(L, dt)
    s.NLayers = int(rs.choice([6, 10, 15, 15, 15, 22]))
    if rs.rand() < 0.3:
        s.tsurfOutputDepth = float(rs.choice([0.0, 0.015, 0.06, 0.3]))
    p = abi.default_parameters(dt)
    for name, lo, hi in _SCALED:
        setattr(p, name, getattr(p, name) * rs.uniform(lo, hi))
    for name, lo, hi in _SHIFTED:
        setattr(p, name, getattr(p, name) + rs.uniform(lo, hi))
    p.PLimSnow = rs.uniform(0.1, 0.45)
    p.PLimRain = rs.uniform(0.55, 0.9)
    # the derived limits follow the reference driver's expressions (InputParameters.cpp:13-21)
    p.MaxWatmms = p.MaxPormms + p.MaxExtmms
    p.WDampLim = 0.1 * p.MaxPormms; p.WWetLim = 0.9 * p.MaxPormms; p.WWearLim = 0.1 * p.MaxPormms
    relax = rs.rand() < 0.5
    s.use_relaxation = 1 if relax else 0
    initlen = int(rs.choice([1, 1, L // 3, L // 2]))
    if initlen > 1:
        f["tsurfobs"][:, :initlen] = f["tair"][:, :initlen] + rs.uniform(-1, 1)
    ls = []
    for i in range(n):
        li = abi.default_local(); li.InitLenI = initlen
        if relax:
            li.tair_relax = float(f["tair"][i, min(initlen, L - 1)]) + rs.uniform(-2, 2)
            li.VZ_relax = float(rs.uniform(0.5, 6)); li.RH_relax = float(rs.uniform(50, 100))
        ls.append(li)
    return f, s, p, ls


@pytest.mark.parametrize("seed", range(16))
def test_random_parameter_sets_match_the_reference(seed):
    from roadsurf_amd import device
    f, s, p, ls = _case(seed)
    ora, _, _ = oh.run_oracle(_kind(), f, s, p, ls)
    res, _ = device.run_points(f, s, p, ls, chunk=int(np.random.RandomState(seed).choice([0, 97, 240])))
    desc = f"seed {seed}: dt {s.DTSecs} NL {s.NLayers} L {s.SimLen} depth {s.tsurfOutputDepth} relax {s.use_relaxation}"
    for k in oh.F64_OUT:
        assert np.array_equal(res[k], ora[k]), (desc, k, int((res[k] != ora[k]).sum()))
    assert (ora["tsurf"] > -100).all(), desc     # the sets are physical: nothing failed
