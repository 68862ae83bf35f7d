set -e
cd $GRAFT_REPO_ROOT
echo "== ASM_PF (committed form), E=128 S=31"; timeout -k 10 120 python scripts/bench_fused.py 16384 128 31 2>&1 | grep attn_block
echo "== W64 S=31"; timeout -k 10 120 python scripts/bench_fused.py 4096 64 31 2>&1 | grep attn_block
touch moleculardiffusion_mivit_amd/csrc/fused_fwd.hip
MIVIT_EXTRA_HIPCC_FLAGS="-DMIVIT_NO_ASM_PF" timeout -k 10 600 python moleculardiffusion_mivit_amd/csrc/build.py > /dev/null 2>&1
echo "== compiler-visible prefetch, E=128 S=31"; timeout -k 10 120 python scripts/bench_fused.py 16384 128 31 2>&1 | grep attn_block
echo "== W64 S=31"; timeout -k 10 120 python scripts/bench_fused.py 4096 64 31 2>&1 | grep attn_block
