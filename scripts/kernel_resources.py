"""Compiler resource report of the MFMA kernels and the NMS kernels (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, no GPU
needed): file | kernel | VGPRs | AGPRs | scratch B/lane | VGPR spills | occupancy waves/SIMD -> stdout."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tiny-faces-pytorch_amd", "csrc")
FILES = sys.argv[1:] or ["conv_dma_bf16.hip", "conv_dma_f16.hip", "conv_dma.hip", "conv3x3h.hip", "wgrad_group.hip", "wgrad_dma.hip", "wgrad3x3.hip", "wgrad.hip", "conv.hip", "stem_conv.hip", "nms.hip"]
print("# compiler resource report (hipcc -Rpass-analysis=kernel-resource-usage, gfx950) of the MFMA kernels and the NMS kernels")
print("# file | kernel | VGPRs | AGPRs | scratch B/lane | VGPR spills | occupancy waves/SIMD")
for f in FILES:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", os.path.join(CSRC, f), "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
        else:
            cur[k] = v
        if k == "VGPRs Spill":
            if all(x in cur for x in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "Occupancy [waves/SIMD]")):
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                print(f"{f} | {name} | {cur['VGPRs']} | {cur['AGPRs']} | {cur['ScratchSize [bytes/lane]']} | {cur['VGPRs Spill']} | {cur['Occupancy [waves/SIMD]']}")
                cur = {}
