"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output for the flash kernels:  python tools/kernel_resources.py /tmp/flash_res.txt [all]"""
import re, sys
txt = open(sys.argv[1]).read()
show_all = len(sys.argv) > 2
for b in txt.split('Function Name: ')[1:]:
    name = b.split('\n')[0]
    m = re.search(r'flash_kernelILi(\d+)ELi(\d+)ELb(\d)ELb(\d)E(?:Li(\d)E)?', name)  # <KS, MODE, STORE_S, F16[, an experiment's extra parameter]>
    if not m:
        continue
    ks, mode, st, f16 = map(int, m.groups()[:4])
    xs = int(m.group(5)) if m.group(5) else 1
    if not show_all and xs == 1 and not (ks == 7 and mode in (2, 3)):
        continue
    def g(k):
        return re.search(k + r': (\d+)', b).group(1)
    print("KS=%d MODE=%d F16=%d XS=%d VGPR=%s AGPR=%s spill=%s scratch=%s occ=%s LDS=%s" % (
        ks, mode, f16, xs, g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
