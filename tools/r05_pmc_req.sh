mkdir -p gpurun_out/r05
G1="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_CACHE_MISS_sum"
G2="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
G3="TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum"
G4="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_MISS_sum TCC_BUBBLE_sum"
for c in fused conv5 copy; do
  a=""; [ $c != fused ] && a="--case $c"
  bash tools/pmc.sh r05/req_$c "$G1" "$G2" "$G3" "$G4" -- python3 $GRAFT_REPO_ROOT/tools/run_one.py --batch 64 --steps 3 halo_shared=0 $a > gpurun_out/r05/req_$c.txt 2>&1
done
tail -40 gpurun_out/r05/req_fused.txt gpurun_out/r05/req_conv5.txt
