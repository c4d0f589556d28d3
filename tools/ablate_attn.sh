set -e
for n in 0 1 2 3 4; do
  CAREL_EXTRA_FLAGS=-DCAREL_ATTN_ABLATE=$n python -m carel_vae_amd.build --force > /dev/null 2>&1
  TAG="ablate $n" python tools/ablate_attn.py
done
python -m carel_vae_amd.build --force > /dev/null 2>&1
