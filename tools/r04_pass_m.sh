#!/bin/bash
# Round-4 pass M: wide column groups of k_gemm_skinny (5 .. 9 rows in one group for wq|wk|wv and w1|w3): parity + the 9-token chunk rate
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "SKINNY_WIDE or 7b_logits or short_chunks or prompt_continuation or chunks_in_one_pass" > $O/r04m_pytest.txt 2>&1; tail -4 $O/r04m_pytest.txt
timeout 600 python -m pytest tests/test_pipeline.py -x -q -m gpu -k "batched_set" >> $O/r04m_pytest.txt 2>&1; tail -3 $O/r04m_pytest.txt
for v in 1000000 1536; do echo "== LLAMAHIP_SKINNY_WIDE_MIN=$v"; LLAMAHIP_SKINNY_WIDE_MIN=$v timeout 300 python tools/chunk_probe.py 2>&1 | tail -4; done | tee $O/r04m_chunk_probe_ab.txt
