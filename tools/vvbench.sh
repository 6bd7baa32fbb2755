for n in 1 2 3 4 5 8; do
  if [ $n = 5 ]; then L=mom6_amd/libmom6hip.so; else L=variants/libmom6hip_vb$n.so; fi
  echo "VB=$n"; MOM6HIP_LIB_PATH=$PWD/$L timeout -k 10 120 python tools/perf_vertvisc.py || exit 1
done
