# Timing ablations of the attention backward kernel: each variant is built as libcarel_hip_attnN.so (own object directory) and loaded through
# CAREL_HIP_LIB -- the product library is never overwritten with a wrong-result build.
set -e
cd "$(dirname "$0")/.."
for n in 0 1 2 3 4; do
  CAREL_BUILD_TAG=attn$n CAREL_EXTRA_FLAGS=-DCAREL_ATTN_ABLATE=$n python -m carel_vae_amd.build > /dev/null 2>&1
  CAREL_HIP_LIB=$PWD/carel_vae_amd/libcarel_hip_attn$n.so TAG="ablate $n" python tools/ablate_attn.py
done
