"""disco_amd/check_dpp_hazards.py -- the check disco_amd/build.py runs on the device listing of the units that read other lanes' registers through
DPP inside inline asm (csrc/dpp64.h): it must flag both hazards hipcc cannot see there, and accept the sequences the helpers emit."""
import os
import sys

import pytest

from disco_amd import check_dpp_hazards as chk

DPP = '\tv_fmac_f64_dpp v[0:1], -v[2:3], v[8:9] row_newbcast:3 row_mask:0xf bank_mask:0xf'


def run(text):
    return chk.check(text.splitlines())


def test_valu_write_of_the_source_within_two_wait_states():
    n, bad = run('\tv_mul_f64 v[2:3], v[4:5], v[6:7]\n' + DPP)
    assert n == 1 and len(bad) == 1 and 'written 0 wait states earlier' in bad[0][1]
    n, bad = run('\tv_mul_f64 v[2:3], v[4:5], v[6:7]\n\tv_mov_b32_e32 v9, 0\n' + DPP)
    assert len(bad) == 1 and 'written 1 wait states earlier' in bad[0][1]
    assert run('\tv_mul_f64 v[2:3], v[4:5], v[6:7]\n\ts_nop 1\n' + DPP)[1] == []
    assert run('\tv_mul_f64 v[2:3], v[4:5], v[6:7]\n\tv_mov_b32_e32 v9, 0\n\tv_mov_b32_e32 v10, 0\n' + DPP)[1] == []
    assert run('\tv_mul_f64 v[4:5], v[4:5], v[6:7]\n' + DPP)[1] == []                 # another register
    assert run('\tv_mov_b32_e32 v3, 0\n' + DPP)[1] != []                              # half of the pair is enough


def test_exec_write_within_five_wait_states():
    assert run('\ts_or_b64 exec, exec, s[2:3]\n\tv_mov_b32_e32 v9, 0\n\ts_nop 1\n' + DPP)[1] != []
    assert run('\ts_and_saveexec_b64 s[0:1], vcc\n\ts_nop 3\n' + DPP)[1] != []
    assert run('\ts_and_saveexec_b64 s[0:1], vcc\n\ts_nop 4\n' + DPP)[1] == []
    assert run('\tv_cmpx_gt_f64 vcc, v[0:1], v[2:3]\n\ts_nop 1\n' + DPP)[1] != []


def test_branch_target_inside_the_window():
    # the listing's predecessor is harmless, but somebody jumps to the label: the instructions after it must cover 5 wait states
    assert run('\tv_mov_b32_e32 v20, 0\n.LBB0_3:\n\ts_nop 1\n' + DPP)[1] != []
    assert run('\tv_mov_b32_e32 v20, 0\n.LBB0_3:\n\ts_nop 4\n' + DPP)[1] == []
    assert run('.LBB0_3:                                ; =>This Inner Loop Header: Depth=1\n\tv_mov_b64_e32 v[40:41], 0\n\ts_nop 4\n\ts_nop 1\n' + DPP)[1] == []


def test_chain_on_the_accumulator_is_no_hazard():
    assert run(DPP + '\n' + DPP + '\n' + DPP) == (3, [])


def test_kernel_split():
    text = '_Z3fooPd:\n\tv_mul_f64 v[2:3], v[4:5], v[6:7]\n' + DPP + '\n\ts_endpgm\n_Z3barPd:\n' + DPP + '\n\ts_endpgm\n'
    ks = chk.kernels(text)
    assert list(ks) == ['_Z3fooPd', '_Z3barPd']
    assert len(chk.check(ks['_Z3fooPd'])[1]) == 1 and chk.check(ks['_Z3barPd'])[1] == []


def test_the_built_listing_is_clean():
    """When the library was built in this tree the listing of the DPP unit is beside its object: every kernel in it passes."""
    from disco_amd import build
    listing = os.path.join(build.OBJ, 'api_solve_dpp-hip-amdgcn-amd-amdhsa-gfx950.s')
    if not os.path.exists(listing):
        pytest.skip('no device listing in this tree (library not built here)')
    n_dpp = 0
    for name, lines in chk.kernels(open(listing).read()).items():
        n, bad = chk.check(lines, name)
        n_dpp += n
        assert bad == [], (name, bad[:3])
    assert n_dpp > 10000
