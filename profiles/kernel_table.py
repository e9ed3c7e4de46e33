"""VGPRs / scratch / spills of every render_kernel instantiation in the built library (llvm-readelf on the code objects)."""
import re, subprocess, sys
so = sys.argv[1] if len(sys.argv) > 1 else '/root/repo/raytracinginoneweekendincuda_amd/librtow_hip.so'
data = open(so, 'rb').read()
idx = [m.start() for m in re.finditer(b'\x7fELF', data)]
rows = []
for i, st in enumerate(idx[1:]):
    end = idx[i + 2] if i + 2 < len(idx) else len(data)
    open('/tmp/kt.elf', 'wb').write(data[st:end])
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', '/tmp/kt.elf'], capture_output=True, text=True).stdout
    cur = {}
    for line in out.splitlines():
        line = line.strip()
        m = re.match(r'\.(name|private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size):\s*(\S+)', line)
        if m: cur[m.group(1)] = m.group(2)
        if line.startswith('.wavefront_size'):
            n = cur.get('name', '')
            m = re.search(r'render_kernelILi(\d)ENS_12_GLOBAL__N_16TraitsILi(\d)ELb(\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d+)ELb(\d)ELb(\d)', n)
            if m:
                strict, world, comp, rich, waves, media, batch, nested, block, fast, grouped = (int(x) for x in m.groups())
                rows.append((world, comp, rich, media, batch, nested, fast, grouped, block, waves, 'strict' if strict else 'fast',
                             int(cur['vgpr_count']), int(cur['private_segment_fixed_size']), int(cur['vgpr_spill_count'])))
            cur = {}
print("world comp rich media batch nested fast grouped block waves build vgpr scratchB spills")
for r in sorted(rows): print(*r)
