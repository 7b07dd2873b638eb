"""Builds libvilgod_hip.so (HIP kernels + C ABI, gfx950 only) in-tree with hipcc.

    python -m vilgod_amd.build            # incremental
    python -m vilgod_amd.build --force
    python -m vilgod_amd.build --dev      # libvilgod_hip_dev.so: the same sources with -DVG_DEV (ablation / trace entry points
                                          # of tools/dev/vilgod_hip_dev.h and the superseded kernels; tools/ only)

The .so lands next to this file so it travels with the repository snapshot to the GPU box; it
is git-ignored.  No torch involvement: plain `hipcc -shared`.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB = os.path.join(HERE, 'libvilgod_hip.so')
OBJ = os.path.join(HERE, 'csrc', '_obj')

ARCH = 'gfx950'
COMMON = ['-O3', '-fPIC', '-std=c++17', f'--offload-arch={ARCH}', f'-I{INCLUDE}', f'-I{CSRC}',
          '-Wno-unused-result', '-DNDEBUG']
# per-source extra flags.  Parity-critical float code is compiled without FMA contraction so that
# only the FMAs written in the source exist (the CPU oracle is compiled the same way).
SOURCES = {
    'api.hip': [],
    'render.hip': ['-ffp-contract=off'],
    'ground.hip': ['-ffp-contract=off'],
    'cluster.hip': ['-ffp-contract=off'],
    'hdbscan_tree.cpp': ['-ffp-contract=off'],
    'hdbscan_device.hip': ['-ffp-contract=off'],
    'segment.hip': ['-ffp-contract=off'],
    'vit.hip': [],
}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True, dev=False):
    OBJ = os.path.join(HERE, 'csrc', '_obj_dev' if dev else '_obj')
    LIB = os.path.join(HERE, 'libvilgod_hip_dev.so' if dev else 'libvilgod_hip.so')
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.inc'))]
    if dev and os.path.isdir(os.path.join(CSRC, 'dev')):           # the development build's kernels (included by vit.hip under VG_DEV)
        headers += [os.path.join(CSRC, 'dev', f) for f in os.listdir(os.path.join(CSRC, 'dev')) if f.endswith('.inc')]
    headers += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    headers.append(os.path.abspath(__file__))
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(OBJ, src.rsplit('.', 1)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [path] + headers):
            lang = ['-x', 'hip'] if src.endswith('.cpp') else []
            jobs.append((src, [hipcc] + COMMON + (['-DVG_DEV'] if dev else []) + extra + lang + ['-c', path, '-o', obj]))

    def run(job):
        src, cmd = job
        if verbose:
            print('[vilgod_amd.build] hipcc', src, flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', LIB] + objs
        if verbose:
            print('[vilgod_amd.build] link', os.path.basename(LIB), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    return LIB


_ASM_CACHE = {}


def _device_asm(source):
    """device-side assembly text of csrc/<source> with the product flags (one compile per process: the checks below share it)"""
    import tempfile
    if source not in _ASM_CACHE:
        hipcc = _hipcc()
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, source + '.s')
            cmd = [hipcc] + [c for c in COMMON if c != '-fPIC'] + SOURCES[source] + ['-S', '--cuda-device-only', os.path.join(CSRC, source), '-o', out]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f'hipcc -S failed for {source}:\n{r.stderr}')
            _ASM_CACHE[source] = open(out).read()
    return _ASM_CACHE[source]


def check_scratch(source='vit.hip', name_part='k_gemm_f16_pp64'):
    """Kernels of `source` whose mangled name contains `name_part` and that use scratch memory (register spills) -> [(kernel, bytes)].
    The projection GEMM runs with 128 accumulator registers per wave and a 256-register budget: an instantiation that spills keeps
    its spill slots busy inside the tile loop, and one that did (round 3) returned wrong values."""
    import re
    text = _device_asm(source)
    found = []
    for m in re.finditer(r'\.set (\S+)\.private_seg_size, (\d+)', text):
        if name_part in m.group(1) and int(m.group(2)) > 0:
            found.append((m.group(1), int(m.group(2))))
    return found


def check_w4_epilogues():
    """k_gemm_f16_w4 (csrc/vit.hip) leaves the next tile's DMA pieces in flight when its K loop's assembly block ends and lets the next
    block wait for them with counts that allow for the epilogue's stores (csrc/gen_gemm_w4.py ST: vector memory operations retire in
    order).  Those counts are only right if the compiler emits the stores the generator assumes.  Per product instantiation (VAR = 0)
    -> {kernel: (stores behind the assembly block, scratch bytes, `s_waitcnt vmcnt(0)` inside the tile loop)}."""
    import re
    text = _device_asm('vit.hip')
    res = {}
    for m in re.finditer(r'^(_Z13k_gemm_f16_w4ILi(\d)ELi(\d)ELi0E\S*):\s', text, flags=re.M):
        name, epi, ln = m.group(1), int(m.group(2)), int(m.group(3))
        body = text[m.end():text.index('s_endpgm', m.end())]
        tail = body[body.rindex('s_nop 15'):]                      # behind the K loop's assembly block (the loop's top is laid out in front of it)
        scratch = int(re.search(r'\.set ' + re.escape(name) + r'\.private_seg_size, (\d+)', text).group(1))
        # compiler waits inside the tile loop: everything between the first-tile set-up and the final drain
        waits = re.findall(r'^\ts_waitcnt vmcnt\((\d+)\)', body, flags=re.M)       # (the assembly block's own waits are not tab-indented)
        res[(epi, ln)] = {'stores': len(re.findall(r'\bglobal_store_', tail)), 'scratch': scratch, 'compiler_vmcnt': [int(w) for w in waits]}
    return res


def check_isa(sources=None, verbose=False):
    """Guard against a code-generation bug of this ROCm's gfx950 back end (found in round 3, csrc/cluster.hip ClGrid::inf): a
    wave-uniform 64-bit constant whose HIGH half is not zero -- `double x = INFINITY` kept in SGPRs -- can be materialised as
    `s_mov_b64 s[..], 0x7ff0000000000000`.  gfx9 encodes 32-bit literals only: the assembler rejects that line, and the direct
    object emission drops the high half (the kernel then sees 0.0).  Compiles the sources to assembly (device side only) and
    returns the offending lines (empty = clean)."""
    import re
    import tempfile
    hipcc = _hipcc()
    bad = []
    pat = re.compile(r's_mov_b64\s+s\[[0-9:]+\],\s*0x[0-9a-fA-F]{9,}')
    for src, extra in SOURCES.items():
        if not src.endswith('.hip') or (sources is not None and src not in sources):
            continue
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, src + '.s')
            cmd = [hipcc] + [c for c in COMMON if c != '-fPIC'] + extra + ['-S', '--cuda-device-only', os.path.join(CSRC, src), '-o', out]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f'hipcc -S failed for {src}:\n{r.stderr}')
            kernel = None
            for line in open(out):
                if line.startswith('_Z') and line.rstrip().endswith(':'):
                    kernel = line.strip()[:-1]
                if pat.search(line):
                    bad.append((src, kernel, line.strip()))
        if verbose:
            print(f'[vilgod_amd.build] isa check {src}: {sum(1 for b in bad if b[0] == src)} unencodable literals', flush=True)
    return bad


if __name__ == '__main__':
    if '--check-isa' in sys.argv:
        found = check_isa(verbose=True)
        for f in found:
            print(*f)
        sys.exit(1 if found else 0)
    print(build(force='--force' in sys.argv, dev='--dev' in sys.argv))
