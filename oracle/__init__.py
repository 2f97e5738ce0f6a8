"""CPU oracle of the hot path: test infrastructure only (see mcgaze_oracle.py, preprocess_oracle.py)."""
