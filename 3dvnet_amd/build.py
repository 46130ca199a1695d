"""Builds lib3dvnet_hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

    python 3dvnet_amd/build.py [--force] [--verbose]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib3dvnet_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-I', os.path.join(ROOT, 'include'), '-I', CSRC] + os.environ.get('V3D_EXTRA_FLAGS', '').split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _flags_tag():
    """Objects are keyed by the compiler flags: a build with V3D_EXTRA_FLAGS (the developer scripts' instrumented /
    ablated variants) lands in its own object directory and can never be mistaken for the default build."""
    return hashlib.sha256(' '.join([HIPCC] + FLAGS).encode()).hexdigest()[:12]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, strict=False):
    """Compile every csrc/*.hip to an object (in parallel) and link the shared library."""
    headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(ROOT, 'include', '*.h'))
    tag = _flags_tag()
    objdir = os.path.join(HERE, 'build', tag)
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(HERE, 'build', 'linked_flags')     # which flag set the in-tree .so was linked from
    linked = open(stamp).read().strip() if os.path.exists(stamp) else None
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [HIPCC] + FLAGS + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % src)
        if verbose and out:
            print(out.decode())
    # ISA guard of the fused decoder (3dvnet_amd/isa_check.py: no scratch; LDS read signature pinned per compiler version) on the
    # default build; developer flag sets are not checked.  `strict` (the entry-point build): the guard must be able to run.
    if not os.environ.get('V3D_EXTRA_FLAGS', '').strip() and not os.environ.get('V3D_SKIP_LDS_CHECK'):
        sys.path.insert(0, HERE)
        try:
            import isa_check
            isa_check.check(os.path.join(objdir, 'decoder.o'), strict=strict)
            # the fused backbone kernels: every 16-byte LDS read must be the operand of a matrix instruction
            isa_check.check_wide_lds(os.path.join(objdir, 'irb.o'), 'irb_kernel', strict=strict)
            isa_check.check_wide_lds(os.path.join(objdir, 'fpn.o'), 'fpn_level_kernel', strict=strict)
        finally:
            sys.path.pop(0)
    if force or procs or linked != tag or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        with open(stamp, 'w') as f:
            f.write(tag + '\n')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv))
