# repeats tools/ipc_soak.py's failing configuration (world 3, N 13312, stratified, eager 8) with a 3 s wait bound
export CSSM_PEER_TIMEOUT_MS=3000
for i in 1 2 3 4 5 6; do
  python tools/ipc_soak.py 3 400 2 8 2>&1 | grep "round\|wait code\|SOAK" | cut -c1-400
done
