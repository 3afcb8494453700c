cd "$GRAFT_REPO_ROOT"
export FI_HIP_LIB=$PWD/exp_libs/libfi_timing.so NOREF=1
for tr in "0 0" "4 30" "3 20" "6 60"; do
  set -- $tr
  echo "cascade levels: terms $1 ratio $2"
  if [ "$1" = "0" ]; then
    MODES="mgmix3:f64:3:1:1:0:0:1e-7" timeout -k 10 120 python tools/exp.py 2>&1 | grep "ms/step"
  else
    FI_COARSE_TERMS=$1 FI_COARSE_RATIO=$2 MODES="mgmix3:f64:3:1:1:0:0:1e-7" timeout -k 10 120 python tools/exp.py 2>&1 | grep "ms/step"
  fi
done
