"""CPU oracle for the U-RNN hot path -- test infrastructure only (see urnn_oracle.c header)."""
