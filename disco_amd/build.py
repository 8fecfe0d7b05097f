"""Build libdisco_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repository snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'disco_hip.hip')
OUT = os.path.join(HERE, 'lib', 'libdisco_hip.so')
DEPS = [os.path.join(HERE, 'csrc', f) for f in os.listdir(os.path.join(HERE, 'csrc'))] + \
       [os.path.join(os.path.dirname(HERE), 'include', 'disco_hip.h')]


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def build_hip(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # -fno-slp-vectorize: hipcc's SLP pass fuses adjacent f32 ops into v_pk_*_f32, which issue slower than the scalar
    # pair on gfx950 (guide: "packed f32 VALU ... an anti-lever"); measured -0.9 ms on k_stft_cov, -0.6 ms on k_step2_cov_fused
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', '-shared', '-fPIC', '-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd), flush=True)
    # the resource-usage remarks go to stderr: k_room_cov_dma counts its vector-memory queue by hand (csrc/k_room.h) and a
    # compiler spill inside its loop would shift that count -- a build whose LDS-DMA kernels use scratch is refused
    p = subprocess.run(cmd + ['-Rpass-analysis=kernel-resource-usage'], stderr=subprocess.PIPE, text=True)
    spilled = scratch_users(p.stderr, 'k_room_cov_dma')
    if p.returncode != 0 or spilled:
        sys.stderr.write(p.stderr[-8000:] if p.returncode != 0 else '')
        if os.path.exists(OUT) and spilled:
            os.remove(OUT)
        raise RuntimeError(f'hipcc failed ({p.returncode})' if p.returncode != 0 else f'kernels that must not spill use scratch: {spilled}')
    return OUT


def scratch_users(remarks, name_part):
    """Names of kernels containing `name_part` whose 'ScratchSize [bytes/lane]' remark is not 0."""
    bad, cur = [], None
    for line in remarks.splitlines():
        if 'Function Name:' in line:
            cur = line.split('Function Name:')[1].split('[')[0].strip()
        elif 'ScratchSize [bytes/lane]:' in line and cur and name_part in cur:
            if int(line.split('ScratchSize [bytes/lane]:')[1].split('[')[0].strip()) != 0:
                bad.append(cur)
    return bad


if __name__ == '__main__':
    build_hip(force='--force' in sys.argv)
