"""Static check of the built gfx950 code objects (no GPU): no kernel may both spill to scratch and be able to run two of its workgroups
on one CU.  Found the hard way on MI355X ("scratch and co-resident workgroups"): the 128 x 128 weight-gradient kernel with
12 bytes of spill per lane returned wrong sums in a third of its launches whenever two working workgroups shared a CU, and never with one
per CU or without the spill.  A kernel that spills is accepted only if ONE workgroup already takes the register file of a CU
(waves x registers per lane == 8 x 256, or more than 256 registers per lane at 4 waves)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def _kernels(tmp_path):
    so = os.path.join(ROOT, 'stair_amd', 'lib', 'libstair_hip.so')
    if not (os.path.exists(so) and os.path.exists(os.path.join(LLVM, 'llvm-objdump'))):
        pytest.skip('library or LLVM tools not present')
    local = str(tmp_path / 'lib.so')
    shutil.copy(so, local)
    subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = []
    for f in sorted(os.listdir(tmp_path)):
        if 'gfx950' not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        for block in re.split(r'\n\s+- \.agpr_count:', notes)[1:]:
            get = lambda key: re.search(r'\.%s:\s+(\S+)' % key, block)
            name = get('name')
            if not name or not get('private_segment_fixed_size'):
                continue
            out.append(dict(name=name.group(1), scratch=int(get('private_segment_fixed_size').group(1)), threads=int(get('max_flat_workgroup_size').group(1)),
                            vgpr=int(get('vgpr_count').group(1)), agpr=int(re.match(r'\s*(\d+)', block).group(1))))
    assert len(out) > 100, len(out)
    return out


def test_no_kernel_spills_while_two_workgroups_can_share_a_cu(tmp_path):
    bad = []
    for k in _kernels(tmp_path):
        if k['scratch'] == 0:
            continue
        regs = k['vgpr'] + k['agpr']                                 # unified file: 512 per SIMD lane, 4 SIMDs per CU
        waves_per_simd = (k['threads'] // 64 + 3) // 4
        one_per_cu = waves_per_simd * max(regs, 1) > 256            # a second workgroup's waves would not fit beside the first's
        if not one_per_cu:
            bad.append((k['name'], k['scratch'], k['threads'], regs))
    assert not bad, bad
