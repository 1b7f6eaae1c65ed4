#!/bin/bash
# r04: which Winograd configuration keeps every gradient of the B=32 step inside the App.-C bars?  (headline fp64 test, worst ratios printed)
cd "$GRAFT_REPO_ROOT" || exit 1
for e in "$@"; do
  echo "=== $e"
  env $e timeout 900 python -m pytest tests/test_headline_gpu.py -q -s -k "train_step_at_the_headline or eval_forward" 2>&1 | grep -E "worst five|passed|failed|gradients off|^E  .*encoder|headline" | cut -c1-600
done
