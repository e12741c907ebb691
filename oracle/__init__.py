"""CPU oracle (test infrastructure only - see vae_oracle.py header)."""
