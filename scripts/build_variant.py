#!/usr/bin/env python3
"""Developer: one source compiled with extra flags and linked with the default objects of all the others into
3dvnet_amd/build/ablate/lib_<tag>.so (scripts that accept V3D_LIB_OVERRIDE load it instead of the default library).
    python scripts/build_variant.py costreg.hip l9a -DV3D_L9_CFG=kDeconvS2,16,8,4,8,28,16,4"""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
out = os.path.join(root, '3dvnet_amd', 'build', 'ablate'); os.makedirs(out, exist_ok=True)
subprocess.check_call([sys.executable, os.path.join(root, '3dvnet_amd', 'build.py')], stdout=subprocess.DEVNULL)
deftag = open(os.path.join(root, '3dvnet_amd', 'build', 'linked_flags')).read().strip()
stem = os.path.splitext(src)[0]
others = [o for o in glob.glob(os.path.join(root, '3dvnet_amd', 'build', deftag, '*.o')) if os.path.basename(o) != stem + '.o']
obj = os.path.join(out, '%s_%s.o' % (stem, tag))
base = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', os.path.join(root, 'include'), '-I', os.path.join(root, '3dvnet_amd', 'csrc')]
subprocess.check_call(base + flags + ['-c', os.path.join(root, '3dvnet_amd', 'csrc', src), '-o', obj])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, 'lib_%s.so' % tag)] + others + [obj])
print('built', tag)
