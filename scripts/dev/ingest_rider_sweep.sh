#!/bin/bash
# where the next batch's PCIe pull rides (ingest.IngestPipeline): parts = carrier launches that share the copy, skip = carriers that go without first
for cfg in "1 0" "1 1" "1 2" "2 0" "2 1" "3 0"; do set -- $cfg; echo "parts $1 skip $2"
  TSGNN_INGEST_PULL_PARTS=$1 TSGNN_INGEST_PULL_SKIP=$2 python scripts/dev/ingest_rider_step.py 2>&1 | grep -E "slot step|new batch"; done
