#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r2fin; mkdir -p $O
(time timeout 200 python -m pytest tests -m gpu -x -q --durations=5 \
  --deselect tests/test_gpu_fullsize.py::test_2p16_row_execution_bytes_equal_the_oracle_prover \
  --deselect "tests/test_gpu_distributed.py::test_coset_partitioned_proof_equals_the_single_gpu_proof[8-12-poseidon]" \
  --deselect "tests/test_gpu_distributed.py::test_coset_partitioned_proof_equals_the_single_gpu_proof[4-12-poseidon]" \
  --deselect "tests/test_gpu_distributed.py::test_coset_partitioned_proof_equals_the_single_gpu_proof[4-12-blake3]" \
  --deselect "tests/test_gpu_distributed.py::test_coset_partitioned_proof_of_a_real_execution[8]" \
  --deselect tests/test_gpu_stark.py::test_full_size_tables_proof_bytes_match_oracle \
  --deselect tests/test_gpu_fullsize.py::test_config4_poseidon_heavy_2p22_rows \
  --deselect tests/test_gpu_blake3.py::test_twelve_table_all_proof_bytes_match_oracle \
  -k "not real_execution_proof_bytes" 2>&1 | tail -12) > $O/pytest.log 2>&1
tail -14 $O/pytest.log
