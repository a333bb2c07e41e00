export TMPDIR=/tmp
NM_GP_CELL=LSTM timeout 600 python tools/general_path_probe.py NM_LSTM_CLUSTER 1 0 2>&1 | grep "NM_LSTM" | tail -3
