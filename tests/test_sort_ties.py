"""Does the order of equal sort keys matter on the BASELINE configs?  (round-2 verdict, "parity unpinned" half 2.)

The reference sorts 64-bit keys (tile << 32 | depth bits) with a library sortperm! (AcceleratedKernels / CUDA.jl,
rasterizer.jl:357-366) and leaves the order of EQUAL keys to that library; this build fixes ascending Gaussian id.  For each
single-GPU BASELINE config this test counts the equal (tile, depth-bits) pairs in the oracle's key list, and renders the
tiles that contain one with BOTH extreme tie orders (ascending and descending id inside every run of equal keys): the
image difference must stay inside the parity tolerance of SURVEY.md §8(c) (|Δ| <= 1e-4 on >= 99.99 % of values), i.e. the
tie rule alone cannot fail a reference comparison at that tolerance (it can use up to half of the outlier budget at
config 5: ties are NOT absent — float32 depths collide inside a tile a few hundred times per view).  CPU only (oracle), ~1 min for all four."""
import numpy as np
import pytest

CONFIGS = [  # name, N, W, H, SH degree, seed (seed = 1000 + config index)
    ("config1", 10_000, 640, 480, 0, 1001),
    ("config2", 100_000, 1920, 1080, 3, 1002),
    ("config3", 1_000_000, 1920, 1080, 3, 1003),
    ("config5", 5_000_000, 3840, 2160, 3, 1005),
]


def tie_report(orc, pkg, n, W, H, deg, seed):
    s = pkg.synthetic.make_scene(n, W, H, deg, seed)
    cam = orc.Camera(W, H, s.focal)
    grid = cam.grid
    depths, radii, means2d, conics, _ = orc.project(s.means, s.scales, s.rotations, cam)
    tiles = orc.count_tiles(means2d, radii, grid)
    offsets, d = orc.cumsum(tiles)
    ku, vu = orc.duplicate_with_keys(means2d, depths, offsets, radii, grid, d)
    ks, vs = orc.sort_pairs(ku, vu)          # stable: ascending id inside a run of equal keys
    eq = ks[1:] == ks[:-1]
    n_pairs = int(eq.sum())
    rep = {"D": int(d), "tied_pairs": n_pairs, "tiles_with_ties": 0, "values_differing": 0, "max_abs_diff": 0.0,
           "frac_over_1e-4": 0.0}
    if n_pairs == 0:
        return rep
    # the other extreme: descending id inside every run of equal keys
    vs_rev = vs.copy()
    idx = np.flatnonzero(eq)
    run_start = idx[np.r_[True, np.diff(idx) > 1]]
    run_end = idx[np.r_[np.diff(idx) > 1, True]] + 1      # inclusive last element of the run
    for a, b in zip(run_start, run_end):
        vs_rev[a:b + 1] = vs[a:b + 1][::-1]
    tie_tiles = np.unique((ks[idx] >> np.uint64(32)).astype(np.int64))
    rep["tiles_with_ties"] = int(tie_tiles.size)
    ranges = orc.identify_tile_range(ks, grid[0] * grid[1])
    only = np.zeros_like(ranges)
    only[tie_tiles] = ranges[tie_tiles]      # render only the tiles that hold a tie; every other pixel: empty list
    rgbs, _ = orc.sh_forward(radii, s.means, cam.camera_center, s.shs, deg)
    bg = np.zeros(3, np.float32)
    opac = s.opacities.reshape(-1)
    img_a = orc.render(W, H, 3, vs, means2d, opac, conics, rgbs, only, bg, n=n)[0]
    img_b = orc.render(W, H, 3, vs_rev, means2d, opac, conics, rgbs, only, bg, n=n)[0]
    diff = np.abs(img_a - img_b)
    rep["values_differing"] = int((diff > 0).sum())
    rep["max_abs_diff"] = float(diff.max())
    rep["frac_over_1e-4"] = float((diff > 1e-4).sum() / diff.size)   # against ALL values of the image, as §8(c) counts
    return rep


@pytest.mark.parametrize("name,n,W,H,deg,seed", CONFIGS)
def test_tie_order_is_immaterial_at_the_parity_tolerance(orc, pkg, name, n, W, H, deg, seed):
    rep = tie_report(orc, pkg, n, W, H, deg, seed)
    print(f"[sort ties] {name}: D = {rep['D']}, equal (tile, depth-bits) pairs = {rep['tied_pairs']} in "
          f"{rep['tiles_with_ties']} tiles; ascending-id vs descending-id order: {rep['values_differing']} image values "
          f"differ, max |Δ| = {rep['max_abs_diff']:.3e}, fraction over 1e-4 = {rep['frac_over_1e-4']:.2e}")
    # observed: 0 / 4 / 118 / 793 tied pairs and 0 / 0 / 8.8e-6 / 5.5e-5 of the image values beyond 1e-4 for
    # configs 1 / 2 / 3 / 5 — inside the 1e-4 outlier budget of the image criterion even for the worst-case order
    assert rep["frac_over_1e-4"] <= 1e-4, rep
    assert rep["tied_pairs"] <= 1e-3 * rep["D"], rep
