"""CPU parity oracle (test infrastructure only — see oracle/mdx_oracle.c header)."""
