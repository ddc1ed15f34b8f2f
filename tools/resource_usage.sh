#!/bin/bash
# profiles/r06_resource_usage.txt: the compiler's resource report and the code size of the 4-SPS kernels (no GPU needed).
#   bash tools/resource_usage.sh > profiles/r06_resource_usage.txt
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
echo "# hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage osmo_trx_amd/csrc/trx_kernel4.hip (the tree of this commit; tools/resource_usage.sh)"
echo "# nb_pull4_kernel = the normal-burst kernel; burst_pull4_kernel<false,false,true,LIST> = the general kernel's common instantiation (LIST: behind the normal-burst kernel)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -I$R/include -c --cuda-device-only --no-gpu-bundle-output \
	-Rpass-analysis=kernel-resource-usage -o $T/k4.o $R/osmo_trx_amd/csrc/trx_kernel4.hip 2>&1 |
	grep "remark:" | sed 's/.*remark: //; s/ \[-Rpass-analysis=kernel-resource-usage\]//' | grep -v "^\s*$" |
	awk '/Function Name/{on = ($0 ~ /pull4/)} on{print}'
echo
echo "# code size (llvm-readelf -s, bytes):"
/opt/rocm/lib/llvm/bin/llvm-readelf -s $T/k4.o | awk '$4 == "FUNC" && $8 ~ /pull4/ {print $3, $8}'
rm -rf $T
