"""Build libdisco_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repository snapshot).

The host side of the C ABI is a handful of translation units (csrc/api_*.hip, one per kernel family); they are compiled in
parallel and linked into the one shared library.  A unit is recompiled only when it or a header changed."""
import hashlib
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'lib', 'libdisco_hip.so')
OBJ = os.path.join(HERE, 'lib', 'obj')
# -fno-slp-vectorize: hipcc's SLP pass fuses adjacent f32 ops into v_pk_*_f32, which issue slower than the scalar
# pair on gfx950 (guide: "packed f32 VALU ... an anti-lever"); measured -0.9 ms on k_stft_cov, -0.6 ms on k_step2_cov_fused
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', '-fPIC']
# scratch (bytes per lane) a kernel may use; a build that exceeds it is refused.  k_gevd_mwf_r1_dpp: none (a spill between its inline-asm
# DPP statements would break the hazard spacing the build checks).  k_room_cov_dma (csrc/k_room.h): its LDS-DMA loads are inline asm, but
# every wait is a full `s_waitcnt vmcnt(0)` (round 4: no hand-counted queue depth any more), so a scratch access cannot break it; 168
# registers at 3 waves / SIMD leave hipcc a few words short in the A-role loop of some shapes (the loader's per-lane constants: <= 5 reloads
# per iteration) -- tolerated up to 64 bytes, anything larger means the accumulators went to scratch and is refused.
SCRATCH_LIMIT = {'k_room_cov_dma': 64, 'k_gevd_mwf_r1_dpp': 0, 'k_apply_istft_wide': 0}
# units whose kernels read other lanes' registers through DPP inside inline asm (csrc/dpp64.h): hipcc cannot see those reads, so
# the two DPP hazards (a VALU write of the source within 2 wait states, an EXEC write within 5) are checked on the device
# assembly of the unit (disco_amd/check_dpp_hazards.py) and a build with a hazard is refused
DPP_UNITS = ('api_solve_dpp',)


def units():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith('api_') and f.endswith('.hip'))


def headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + \
        [os.path.join(os.path.dirname(HERE), 'include', 'disco_hip.h')]


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


FLAG_STAMP = OUT + '.flags'     # the extra flags (DISCO_CXXFLAGS) the library at OUT was linked from: a variant build must not pass for the default one


def _stamp():
    return open(FLAG_STAMP).read() if os.path.exists(FLAG_STAMP) else ''


def up_to_date(extra=()):
    return _newer(OUT, units() + headers()) and _stamp() == ' '.join(extra)


def _extra_flags():
    """-D switches for A/B builds (DISCO_CXXFLAGS='-DDISCO_X=1 ...'); part of an object's identity."""
    return os.environ.get('DISCO_CXXFLAGS', '').split()


def _compile(src, hipcc, extra, verbose):
    tag = hashlib.sha1(' '.join(extra).encode()).hexdigest()[:8] if extra else 'default'
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.' + tag + '.o')
    if _newer(obj, [src] + headers()):
        return obj, '', 0
    cmd = [hipcc] + FLAGS + extra + ['-c', '-o', obj, src, '-Rpass-analysis=kernel-resource-usage']
    unit = os.path.basename(src)[:-4]
    if unit in DPP_UNITS:
        cmd.insert(-1, '--save-temps=obj')
    if verbose:
        print(' '.join(cmd[:-1]), flush=True)
    t0 = time.time()
    p = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0 and os.path.exists(obj):
        os.remove(obj)
    if p.returncode == 0:
        open(obj[:-2] + '.remarks', 'w').write(p.stderr)          # registers / scratch / LDS / occupancy of every kernel of the unit
    if p.returncode == 0 and unit in DPP_UNITS:
        hazards = _dpp_hazards(unit)
        if hazards:
            os.remove(obj)
            return obj, 'DPP hazards in the generated code:\n' + '\n'.join(hazards[:20]) + '\n', 1
    if verbose:
        print(f'  {os.path.basename(src)}: {time.time() - t0:.0f} s', flush=True)
    return obj, p.stderr, p.returncode


def _dpp_hazards(unit):
    """Runs check_dpp_hazards.py's check over the device listing --save-temps left beside the object; removes the other temporaries."""
    from disco_amd import check_dpp_hazards as chk
    listing = os.path.join(OBJ, unit + '-hip-amdgcn-amd-amdhsa-gfx950.s')
    out, n_dpp = [], 0
    for name, lines in chk.kernels(open(listing).read()).items():
        n, bad = chk.check(lines, name)
        n_dpp += n
        out += [f'{name} +{ln}: {msg}' for ln, msg in bad]
    if n_dpp == 0:
        out.append(f'{listing}: no DPP instruction found (listing not understood?)')
    for f in os.listdir(OBJ):
        if (f.startswith(unit + '-') or f.startswith(unit + '.hip-')) and f != os.path.basename(listing):
            os.remove(os.path.join(OBJ, f))
    return out


def build_hip(force=False, verbose=True, jobs=None):
    extra = _extra_flags()
    if not force and up_to_date(extra):
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    jobs = jobs or int(os.environ.get('DISCO_BUILD_JOBS', os.cpu_count() or 4))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda s: _compile(s, hipcc, extra, verbose), units()))
    objs = []
    for obj, remarks, rc in results:
        if rc != 0:
            sys.stderr.write(remarks[-8000:])
            raise RuntimeError(f'hipcc failed ({rc}) on {obj}')
        spilled = [k for part, limit in SCRATCH_LIMIT.items() for k in scratch_users(remarks, part, limit)]
        if spilled:
            os.remove(obj)
            raise RuntimeError(f'kernels over their scratch allowance: {spilled}')
        objs.append(obj)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(FLAG_STAMP, 'w').write(' '.join(extra))
    return OUT


def scratch_users(remarks, name_part, limit=0):
    """Names (with the size) of kernels containing `name_part` whose 'ScratchSize [bytes/lane]' remark exceeds `limit`."""
    bad, cur = [], None
    for line in remarks.splitlines():
        if 'Function Name:' in line:
            cur = line.split('Function Name:')[1].split('[')[0].strip()
        elif 'ScratchSize [bytes/lane]:' in line and cur and name_part in cur:
            n = int(line.split('ScratchSize [bytes/lane]:')[1].split('[')[0].strip())
            if n > limit:
                bad.append(f'{cur} ({n} B)')
    return bad


if __name__ == '__main__':
    build_hip(force='--force' in sys.argv)
