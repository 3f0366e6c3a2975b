"""No kernel of the default float32 path spills to scratch: the code-object metadata of the built translation
units says `private_segment_fixed_size: 0` for every kernel the dispatcher can pick for the BASELINE configurations and the
bench's side workloads -- the component-parallel kernels (all modes, both integrators), the float32 forward kernels, the multi-wave
record kernels of bodies up to 256 points, the splat / loss / staging kernels.  A scratch access is a vector-memory operation in the
wave's in-order `vmcnt` queue: in loops whose waits are counted (`s_waitcnt vmcnt(k)`) it stalls everything behind it."""
import os
import sys

import pytest

from tests.conftest import REPO

sys.path.insert(0, os.path.join(REPO, 'tools'))

DEFAULT_PATH_OBJECTS = {
    'rollout_fwd_cp_fast.o', 'rollout_bwd_cp_fast.o', 'rollout_bwd_cp_stream_fast.o', 'rollout_bwd_dyn_cp_fast.o', 'rollout_bwd_dyn_cp_stream_fast.o',
    'rollout_bwd_xs_fast.o', 'rollout_bwd_xs_win_fast.o', 'rollout_fwd_fast.o', 'rollout_fwd_split_fast.o', 'rollout_fwd_zmu_fast.o', 'rollout_fwd_cost.o', 'rollout_bwd_mw_fast.o',
    'bev_splat.o', 'physics_loss.o', 'terrain_stage.o', 'heightmap.o', 'interp_grid_fast.o',
}


@pytest.fixture(scope='module')
def metadata():
    import __graft_entry__ as g
    g.build()
    import kernel_metadata
    return kernel_metadata.kernels()


def test_default_path_kernels_use_no_scratch(metadata):
    seen = {o for o, _, _ in metadata}
    assert DEFAULT_PATH_OBJECTS <= seen, sorted(DEFAULT_PATH_OBJECTS - seen)
    bad = []
    for o, name, m in metadata:
        if o not in DEFAULT_PATH_OBJECTS:
            continue
        # one rollout of 257..512 points over EIGHT waves: 512-thread workgroups are held to 128 registers per lane; documented
        # (DESIGN.md 8), not a BASELINE shape (the reference's robots have 175 / 223 points: four waves)
        if 'rollout_bwd_mw_kernel<float, 512' in name or 'rollout_fwd_kernel<float, 512' in name:
            continue
        if m['scratch']:
            bad.append((o, name, m['scratch']))
    assert not bad, bad


def test_streaming_backward_fits_two_workgroups_per_cu(metadata):
    """The 6-slot positions-only streaming kernels run two workgroups (six waves) per CU: <= 256 registers and <= 80 KB of LDS each."""
    rows = [(n, m) for o, n, m in metadata if o == 'rollout_bwd_cp_stream_fast.o' and 'true, true, 3, 6' in n or o == 'rollout_bwd_cp_stream_fast.o' and 'true, false, 3, 6' in n]
    assert len(rows) == 2, rows
    for n, m in rows:
        assert m['vgpr'] + m['agpr'] <= 256 and m['lds'] <= 80 * 1024 and m['scratch'] == 0, (n, m)


def test_lds_window_kernels_fit_one_workgroup_of_eight_waves_per_cu(metadata):
    """The LDS-window backward kernels (round 5) hold a 128 x 128-cell window of both gradient maps (128 KB of the CU's 160 KB: one workgroup
    per CU) and run as workgroups of up to eight waves -- two per SIMD: at most 256 registers per lane, no scratch."""
    rows = [(n, m) for o, n, m in metadata if 2 * 128 * 128 * 4 <= m['lds'] <= 2 * 128 * 128 * 4 + 256]      # (+ the fused loss's eight wave sums)
    names = ' '.join(n for n, _ in rows)
    assert 'rollout_bwd_kernel<float, 4, 1, 1, true, false, true, true, true, true, false>' in names       # positions-only, interleaved maps, carry-over
    assert 'rollout_bwd_kernel<float, 4, 1, 1, true, false, true, true, true, true, true>' in names        # ... with the fused physics loss (round 6)
    assert 'rollout_bwd_kernel<float, 4, 1, 0, true, false, false, true, false, true, false>' in names     # dynamics(), plain maps, no carry-over
    assert 'rollout_bwd_cp_kernel<float, 1, true, false, 0, 6, 3, false, true, false>' in names     # component-parallel early recompute
    assert 'rollout_bwd_cp_kernel<float, 1, true, false, 0, 6, 3, false, true, true>' in names      # ... with the fused physics loss (round 6)
    assert len(rows) >= 20, names
    for n, m in rows:
        assert m['scratch'] == 0 and m['vgpr'] + m['agpr'] <= 256, (n, m)
