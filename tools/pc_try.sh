for cfg in "ABFT_HIP_PANEL_CHUNK=8" "ABFT_HIP_PANEL_LAG=1000" "ABFT_HIP_PANEL_LAG=2" "ABFT_HIP_PANEL_CHUNK=4" "ABFT_HIP_PANEL_CHUNK=2"; do
  echo "--- $cfg"
  env $cfg ABFT_HIP_COO_PC=1 timeout -k 5 120 python3 bench.py --cpu-iters 0 --no-probe --no-extras --steps 10 --warmup 2 --fmt coo --mode sec7 --spec powerlaw:2097152,2 2>&1 | tail -c 300
  echo
done
