# List every esr:: kernel that uses scratch memory (register spills or dynamically indexed private arrays), from hipcc's
# -Rpass-analysis=kernel-resource-usage remarks.  Runs on the build box (no GPU needed).  Round 2 found an 80-VGPR spill
# in the filtered epilogue of score_gemm_kernel<2, false> this way, after the PMC pass showed 20x the expected writes.
cd esrecsys_amd/csrc
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -I../../include $f -o /tmp/spillcheck.o \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|ScratchSize|VGPRs Spill" | paste - - - |
    sed 's/remark: [^ ]* //g; s/\[-Rpass[^]]*\]//g' | grep -v rocprim |
    awk -v f=$f '{ if ($0 !~ /ScratchSize \[bytes\/lane\]: 0 /) print f": "$0 }' | cut -c1-200
done
