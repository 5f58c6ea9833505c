"""Disassemble the gfx950 code object(s) of a HIP object / shared object: python tools/disasm.py <file.o|.so> <out.s>  (profiling aid)"""
import subprocess, sys, tempfile, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_abi import _gfx950_code_objects
outs = []
for i, co in enumerate(_gfx950_code_objects(sys.argv[1])):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(co); f.flush()
        outs.append(subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '-d', '--mcpu=gfx950', f.name], capture_output=True, text=True, check=True).stdout)
open(sys.argv[2], 'w').write('\n'.join(outs))
