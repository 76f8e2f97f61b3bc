"""Static guard for the packed-fp32 cross-wave disturbance (profiles/r03_packed_fp32_hazard.md).

On MI355X a `v_pk_{mul,fma,add}_f32` whose LOW result takes the HIGH register of a source pair (op_sel) computes with that operand
read as zero in lanes 48-63 now and then, while a wave of `gather_conv_split_kernel` shares its SIMD (stand-alone reproducer:
profiles/ub/pk_hazard.hip).  The in-tree kernels run beside that kernel on other streams, so NONE of them may contain a packed-fp32
instruction: csrc/build.sh switches the target feature off; this test disassembles the built library and holds it to that."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "animatablegaussians_amd", "lib", "libag_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PACKED_FP32 = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_library_contains_no_packed_fp32_instructions():
    with tempfile.TemporaryDirectory() as tmp:
        lib = shutil.copy(LIB, os.path.join(tmp, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", lib], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f]
        assert objs, "no gfx950 code objects found in the library"
        kernels = bad = 0
        offenders = set()
        for o in objs:
            dis = subprocess.run([OBJDUMP, "-d", o], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = m.group(1)
                    kernels += 1
                elif PACKED_FP32.search(line):
                    bad += 1
                    offenders.add(cur)
        assert kernels > 50, kernels                    # the disassembly really covered the library's kernels
        assert bad == 0, f"{bad} packed-fp32 instructions in {sorted(offenders)[:8]}: build.sh must keep -packed-fp32-ops off"
