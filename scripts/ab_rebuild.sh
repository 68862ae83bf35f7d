# A/B of a compile-time switch on the GPU box: ab_rebuild.sh <source.hip> "<extra hipcc flags>" <command ...>
# runs the command on the library as shipped, rebuilds with the extra flags (objects of <source.hip> only), runs it again
set -e
cd $GRAFT_REPO_ROOT
SRC=$1; FLAGS=$2; shift 2
echo "== as shipped"; "$@"
touch moleculardiffusion_mivit_amd/csrc/$SRC
MIVIT_EXTRA_HIPCC_FLAGS="$FLAGS" timeout -k 10 600 python moleculardiffusion_mivit_amd/csrc/build.py > /dev/null 2>&1
echo "== rebuilt with $FLAGS"; "$@"
