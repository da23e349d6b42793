cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_ticket; mkdir -p $O
for i in 1 2; do
  IMT_GEMM_LN_TICKET=0 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --breakdown > $O/code_$i.json 2> $O/code_$i.txt || exit 1
  echo "ticket code compiled in, switched off: $(grep 'bench\] gpu' $O/code_$i.txt) $(grep 'gemm_ws_bf16_nt ' $O/code_$i.txt | awk '{print $3}')"
  IMT_LIB=$PWD/build/libimt_hip_noticket.so python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --breakdown > $O/nocode_$i.json 2> $O/nocode_$i.txt || exit 1
  echo "no ticket code: $(grep 'bench\] gpu' $O/nocode_$i.txt) $(grep 'gemm_ws_bf16_nt ' $O/nocode_$i.txt | awk '{print $3}')"
done
