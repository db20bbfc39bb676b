#!/bin/bash
# gfx950 disassembly of one translation unit of the built library:  tools/disasm.sh msm_var > /tmp/msm_var.s
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin octopuszk_amd/_obj/$1.hip.o $T/fb.bin
$L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fb.bin --output=$T/x.co
$L/llvm-objdump -d --no-show-raw-insn $T/x.co | c++filt
rm -rf $T
