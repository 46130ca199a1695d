"""Build-time ISA guard of the fused hypothesis decoder (``csrc/decoder.hip``).

Two properties of ``decoder_fused_kernel`` are checked on the object the default build just compiled:

1. **No scratch.**  The kernel must not contain a single ``scratch_`` instruction: the round-3/4 kernel spilled 47
   registers (163 MB of scratch writes per launch for 0.2 MB of results, on the vector-memory path that bounded it).
   Compiler-independent; always a hard failure.
2. **LDS read signature.**  Observed on gfx950 / ROCm 7.2 (reproducer: ``scripts/micro/lds_b128.hip``): when hipcc merged
   the consecutive corner-table words a thread reads from LDS into 16-byte reads, lanes 48..63 received stale data whenever
   a second wave on the SIMD had MFMAs in flight.  Nothing in the language stops a compiler from re-forming such a read, so
   the number of LDS reads of every width is pinned for the build that passed the repeated-launch determinism test on
   hardware (``tests/test_parity_net_gpu.py::test_fused_decoder_is_deterministic_under_load``).  The pin is keyed by the
   compiler version: with THAT compiler a moved signature fails the build (an edit of the kernel has to re-run the test on a
   GPU and re-pin, ``python -m 3dvnet_amd.isa_check <decoder.o> --print``); with another compiler the signature is expected to
   move, and the guard prints a loud warning instead (the determinism test is part of the GPU suite either way).

``check(obj, strict)``: ``strict=True`` (``__graft_entry__.build()``, the build that produces the shipped library) also fails
when the guard cannot run at all (llvm binutils missing); ``strict=False`` reports that on stderr.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get('V3D_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
KERNEL = 'decoder_fused_kernel'
# compiler the signature below was verified with on MI355X (first line of `hipcc --version`)
PINNED_COMPILER = 'HIP version: 7.2.26015-fc0010cf6a'
# Verified on MI355X (determinism test: 60 launches beside a GEMM on a second stream, bit-identical; golden and unfused-chain
# comparisons): the 16-byte reads are the A fragments from the weight ring (consumed by matrix instructions) and the bias / head
# constants; everything vector instructions consume inside the matrix loop is read 4 or 8 bytes at a time (the staged B fragment
# through four inline-asm ds_read_b64); the eight ds_read2_b64 are the corner-table reads of the first tile's prologue, where no
# matrix instruction is in flight yet.
PINNED = {'ds_read2_b32': 32, 'ds_read2_b64': 8, 'ds_read_b128': 508, 'ds_read_b32': 34, 'ds_read_b64': 12}


class GuardUnavailable(RuntimeError):
    """The guard could not run (tools missing / object not readable)."""


def compiler_id():
    try:
        return subprocess.check_output([HIPCC, '--version'], text=True, stderr=subprocess.STDOUT).splitlines()[0].strip()
    except (OSError, subprocess.CalledProcessError, IndexError) as e:
        raise GuardUnavailable('cannot query %s --version: %s' % (HIPCC, e))


def disassemble(obj):
    try:
        with tempfile.TemporaryDirectory() as td:
            fat, co = os.path.join(td, 'fat'), os.path.join(td, 'co')
            subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, obj])
            subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + fat,
                                   '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
            return subprocess.check_output([os.path.join(LLVM, 'llvm-objdump'), '-d', co], text=True)
    except (OSError, subprocess.CalledProcessError) as e:
        raise GuardUnavailable('cannot disassemble %s: %s' % (obj, e))


def signature(obj):
    """-> ({ds_read mnemonic: count}, number of scratch instructions) of the fused kernel in `obj`."""
    inside, counts, scratch, seen = False, collections.Counter(), 0, False
    for line in disassemble(obj).splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:', line)
        if m:
            inside = KERNEL in m.group(1)
            seen = seen or inside
            continue
        if inside:
            m = re.search(r'\b(ds_read[0-9a-z_]*)\b', line)
            if m:
                counts[m.group(1)] += 1
            if re.search(r'\bscratch_(load|store)', line):
                scratch += 1
    if not seen:
        raise GuardUnavailable('%s not found in %s' % (KERNEL, obj))
    return dict(sorted(counts.items())), scratch


def check(obj, strict=False):
    try:
        sig, scratch = signature(obj)
        comp = compiler_id()
    except GuardUnavailable as e:
        if strict:
            raise RuntimeError('ISA guard of the fused decoder could not run: %s' % e)
        sys.stderr.write('isa_check: guard NOT run (%s)\n' % e)
        return None
    if scratch:
        raise RuntimeError('%s contains %d scratch instructions: the kernel must not spill (DESIGN.md §8.4)' % (KERNEL, scratch))
    if PINNED is None:
        sys.stderr.write('isa_check: LDS read signature of %s is not pinned yet: %s\n' % (KERNEL, sig))
        return sig
    if sig != PINNED:
        msg = ('LDS read signature of %s moved: found %s, pinned %s.\nA wide LDS read of a self-written table returned stale '
               'lanes on gfx950 (DESIGN.md §8.4).  Re-run tests/test_parity_net_gpu.py -k fused on a GPU with this build, then '
               'update PINNED in 3dvnet_amd/isa_check.py.' % (KERNEL, sig, PINNED))
        if comp == PINNED_COMPILER:
            raise RuntimeError(msg)
        sys.stderr.write('isa_check: WARNING (compiler %r, pinned with %r): %s\n' % (comp, PINNED_COMPILER, msg))
    return sig


# ----------------------------------------------------------------------------------------------------------------------------
# Round 6: the fused backbone kernels (csrc/irb.hip, csrc/fpn.hip) follow the same rule by construction -- every LDS read that
# returns 16 bytes per lane is an operand fragment of a matrix instruction, everything vector instructions consume is read 8
# bytes at a time by hand-written ds_read_b64 -- and the guard checks the RULE rather than a count: in program order, the first
# reader of any register written by a ds_read_b128 / ds_read2_b64 must be a v_mfma.
WIDE_LDS = ('ds_read_b128', 'ds_read2_b64', 'ds_read2st64_b64', 'ds_read_b96')
ALL_SOURCES = ('ds_write', 'global_store', 'buffer_store', 'flat_store', 'scratch_store', 'v_cmp', 'v_cmpx', 's_', 'global_atomic',
               'ds_add', 'ds_max', 'ds_min', 'exp')


def _regs(operand):
    """VGPR numbers named by one operand: v7 -> {7}, v[4:7] -> {4..7}; accumulator registers and everything else -> {}."""
    m = re.fullmatch(r'v(\d+)', operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def wide_lds_consumers(obj, kernel_pattern):
    """-> list of (kernel, wide read, first consumer) where the first consumer is NOT a matrix instruction."""
    bad, inside, name, pending = [], False, None, {}
    for line in disassemble(obj).splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:', line)
        if m:
            inside, name, pending = re.search(kernel_pattern, m.group(1)) is not None, m.group(1), {}
            continue
        if not inside:
            continue
        m = re.match(r'^\s*([a-z][a-z0-9_]*)\s*(.*?)\s*(//.*)?$', line)
        if not m or not m.group(1):
            continue
        mnem = m.group(1)
        ops = [o.strip() for o in re.split(r',\s*(?![^\[]*\])', m.group(2).split(' offset')[0]) if o.strip()]
        ops = [o.split()[0] for o in ops if o.split()]
        all_src = mnem.startswith(ALL_SOURCES)
        srcs = set().union(*[_regs(o) for o in (ops if all_src else ops[1:])]) if ops else set()
        dst = set() if all_src or not ops else _regs(ops[0])
        hit = [pending[r] for r in srcs if r in pending]
        if hit and not mnem.startswith('v_mfma'):
            bad.append((name, hit[0], line.strip()))
        for r in srcs | dst:                       # consumed (by a matrix instruction or reported) or overwritten: no longer tracked
            pending.pop(r, None)
        if mnem in WIDE_LDS:
            for r in dst:
                pending[r] = line.strip()
    return bad


def check_wide_lds(obj, kernel_pattern, strict=False):
    try:
        bad = wide_lds_consumers(obj, kernel_pattern)
    except GuardUnavailable as e:
        if strict:
            raise RuntimeError('ISA guard (%s) could not run: %s' % (kernel_pattern, e))
        sys.stderr.write('isa_check: guard NOT run (%s)\n' % e)
        return None
    if bad:
        raise RuntimeError('a 16-byte LDS read feeds a non-matrix instruction (DESIGN.md §8.4) in %s:\n  %s\n  -> %s\n(%d such reads)'
                           % (bad[0][0], bad[0][1], bad[0][2], len(bad)))
    return 0


if __name__ == '__main__':
    if '--wide' in sys.argv:
        for b in wide_lds_consumers(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 else '.'):
            print(b)
    elif '--print' in sys.argv:
        print(compiler_id())
        print(signature(sys.argv[1]))
    else:
        print('ok', check(sys.argv[1], strict=True))
