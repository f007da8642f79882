"""Trajectory-analysis row: RDF pair pass on the GPU, MSD / VACF on the host, against
golden vectors produced by the reference's own Python module (oracle/make_golden_analysis.py)."""
import numpy as np
import pytest

from conftest import GOLDEN
from ljmd_amd import analysis, io_formats, synthetic


def _oracle_hist(oracle):
    return lambda x, y, z, L, nbins, rmax, hist: oracle.rdf_histogram_np(
        np.ascontiguousarray(x), np.ascontiguousarray(y), np.ascontiguousarray(z), L, nbins, rmax, hist)


def test_rva_reader_and_msd_vacf_match_reference_module(golden):
    g = golden("analysis_n108")
    header, snaps = io_formats.read_rva(GOLDEN / "ref_run_n108_oi100" / "rva.dat")
    assert header["n"] == int(g["n"]) and header["box_length"] == float(g["L"])
    assert snaps.shape[0] == int(g["n_snapshots"])
    # our reader and the reference's read_rva decode the same bytes
    assert np.array_equal(snaps[0, 0, 0], g["rx0"]) and np.array_equal(snaps[-1, 1, 0], g["rux_last"])
    assert np.array_equal(snaps[-1, 2, 0], g["vx_last"])
    ru, v = snaps[:, 1], snaps[:, 2]
    assert np.array_equal(analysis.compute_msd_tau_timeorig(ru[:, 0], ru[:, 1], ru[:, 2]), g["msd"])
    assert np.array_equal(analysis.compute_vacf_tau_timeorig(v[:, 0], v[:, 1], v[:, 2]), g["vacf"])
    assert np.array_equal(analysis.compute_msd_tau_timeorig(ru[:, 0], ru[:, 1], ru[:, 2], max_lag=4, origin_stride=2),
                          g["msd_lag4_stride2"])
    assert analysis.compute_msd_tau_timeorig(ru[:1, 0], ru[:1, 1], ru[:1, 2]).tolist() == [0.0]


def test_rdf_oracle_restatement_matches_reference_module(golden, oracle):
    """CPU: compute_rdf with the numpy pair pass injected == the reference's compute_rdf, bit for bit."""
    g = golden("analysis_n108")
    _, snaps = io_formats.read_rva(GOLDEN / "ref_run_n108_oi100" / "rva.dat")
    r = snaps[:, 0]
    rc, gr = analysis.compute_rdf(r[:, 0], r[:, 1], r[:, 2], float(g["L"]), nbins=200, histogram=_oracle_hist(oracle))
    assert np.array_equal(rc, g["r_centers"]) and np.array_equal(gr, g["g"])
    g2 = golden("analysis_rdf_n1200")
    s = g2["snaps"]
    rc2, gr2 = analysis.compute_rdf(s[:, 0], s[:, 1], s[:, 2], float(g2["L"]), nbins=int(g2["nbins"]),
                                    rmax=float(g2["rmax"]), histogram=_oracle_hist(oracle))
    assert np.array_equal(rc2, g2["r_centers"]) and np.array_equal(gr2, g2["g"])   # n = 1200 > 800: sub-sampled


@pytest.mark.gpu
def test_rdf_gpu_matches_reference_golden(golden):
    g = golden("analysis_n108")
    _, snaps = io_formats.read_rva(GOLDEN / "ref_run_n108_oi100" / "rva.dat")
    r = snaps[:, 0]
    rc, gr = analysis.compute_rdf(r[:, 0], r[:, 1], r[:, 2], float(g["L"]), nbins=200)
    assert np.array_equal(rc, g["r_centers"]) and np.array_equal(gr, g["g"])
    g2 = golden("analysis_rdf_n1200")
    s = g2["snaps"]
    rc2, gr2 = analysis.compute_rdf(s[:, 0], s[:, 1], s[:, 2], float(g2["L"]), nbins=int(g2["nbins"]), rmax=float(g2["rmax"]))
    assert np.array_equal(gr2, g2["g"])


@pytest.mark.gpu
def test_rdf_gpu_histogram_bit_exact_vs_oracle_all_particles(oracle):
    """No sub-sampling, n = 6000 (the oracle's numpy loop takes seconds): integer histograms identical;
    at n = 262144 the counts must add up to the number of ordered pairs inside rmax and g -> 1 at large r."""
    p, r, _ = synthetic.make_config(6000, seed=2)
    L = p.box_length
    h_gpu = np.zeros(300, dtype=np.uint64)
    h_ora = np.zeros(300, dtype=np.uint64)
    analysis.rdf_histogram(r[0], r[1], r[2], L, 300, 0.5 * L, h_gpu)
    oracle.rdf_histogram_np(r[0].copy(), r[1].copy(), r[2].copy(), L, 300, 0.5 * L, h_ora)
    assert np.array_equal(h_gpu, h_ora) and h_gpu.sum() > 0
    n = 262144
    p, r, _ = synthetic.make_config(n)
    rc, g = analysis.compute_rdf(r[None, 0], r[None, 1], r[None, 2], p.box_length, nbins=400, subsample=False)
    assert abs(g[-50:].mean() - 1.0) < 5e-3                      # ideal-gas limit at r ~ L/2 (lattice ripples remain)
    assert g[:9].sum() == 0.0 and g[9:16].sum() > 0.0                # first shell of the jittered lattice at 1.08 sigma


@pytest.mark.gpu
@pytest.mark.parametrize("L,rmax_frac,nbins", [(8.0, 0.5, 40), (10.0, 0.5, 50), (7.3, 0.37, 64)])
def test_rdf_gpu_exact_on_ties_and_bin_edges(oracle, L, rmax_frac, nbins):
    """A perfect 8^3 lattice: coordinate differences are exact multiples of L/8, so d/L hits +-0.5 exactly
    (the minimum-image tie, np.rint's half-to-even) and r/dr lands exactly on bin edges -- the cases where the
    kernel's multiply-instead-of-divide fast paths must hand over to the true divisions.  Plus random points
    snapped to a coarse grid.  Integer histograms must equal the reference's numpy arithmetic exactly."""
    k = 8
    g = (np.arange(k) * (L / k))
    x, y, z = (a.ravel().copy() for a in np.meshgrid(g, g, g, indexing="ij"))
    rng = np.random.Generator(np.random.PCG64(3))
    snapped = np.round(rng.uniform(0, L, size=(3, 700)) * 16) / 16 % L       # multiples of 1/16: more exact ties
    x, y, z = (np.concatenate([a, b]) for a, b in zip((x, y, z), snapped))
    h_gpu = np.zeros(nbins, dtype=np.uint64)
    h_ora = np.zeros(nbins, dtype=np.uint64)
    analysis.rdf_histogram(x, y, z, L, nbins, rmax_frac * L, h_gpu)
    oracle.rdf_histogram_np(x.copy(), y.copy(), z.copy(), L, nbins, rmax_frac * L, h_ora)
    assert np.array_equal(h_gpu, h_ora) and h_gpu.sum() > 0


@pytest.mark.gpu
def test_msd_vacf_gpu_match_reference_golden_and_numpy_at_size(golden):
    """MSD / VACF on the GPU (ljmd_time_origin_average) against the reference module's own output on the reference's
    rva.dat (tests/golden/analysis_n108.npz: 90 snapshots of N = 108) -- equal to rounding, 1e-13 relative: the
    reference's np.mean sums pairwise, the kernel in a fixed strided order -- with max_lag / origin_stride as the
    reference passes them, and against the numpy mirror on a synthetic trajectory of 8192 particles x 24 snapshots."""
    g = golden("analysis_n108")
    _, snaps = io_formats.read_rva(GOLDEN / "ref_run_n108_oi100" / "rva.dat")
    ru, v = snaps[:, 1], snaps[:, 2]

    def close(a, b):
        return a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-13 * max(np.max(np.abs(b)), 1e-300)

    assert close(analysis.time_origin_average_gpu(0, ru[:, 0], ru[:, 1], ru[:, 2]), g["msd"])
    assert close(analysis.time_origin_average_gpu(1, v[:, 0], v[:, 1], v[:, 2]), g["vacf"])
    assert close(analysis.time_origin_average_gpu(0, ru[:, 0], ru[:, 1], ru[:, 2], max_lag=4, origin_stride=2),
                 g["msd_lag4_stride2"])
    rng = np.random.Generator(np.random.PCG64(11))
    walk = np.cumsum(rng.normal(0.0, 0.05, size=(3, 24, 8192)), axis=1) + rng.uniform(0, 20, size=(3, 1, 8192))
    vel = rng.normal(0.0, 1.0, size=(3, 24, 8192))
    for stride, lag in ((1, None), (3, 10)):
        assert close(analysis.time_origin_average_gpu(0, *walk, max_lag=lag, origin_stride=stride),
                     analysis.compute_msd_tau_timeorig(*walk, max_lag=lag, origin_stride=stride))
        assert close(analysis.time_origin_average_gpu(1, *vel, max_lag=lag, origin_stride=stride),
                     analysis.compute_vacf_tau_timeorig(*vel, max_lag=lag, origin_stride=stride))
