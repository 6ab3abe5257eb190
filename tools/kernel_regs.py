import subprocess,re,os,tempfile,shutil,sys
lib=sys.argv[1]; pat=sys.argv[2]
d=tempfile.mkdtemp()
shutil.copy(lib,d+'/lib.so')
subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','--offloading','lib.so'],cwd=d,check=True,capture_output=True)
for f in sorted(os.listdir(d)):
    if 'gfx950' not in f: continue
    notes=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf','--notes',f],cwd=d,capture_output=True,text=True).stdout
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)",notes):
        if re.search(pat,m.group(1)): print(m.group(1)[:66],m.group(2),m.group(3))
