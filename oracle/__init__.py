"""CPU oracle (test infrastructure).  See codec_oracle.py.  Not imported by gan_amd/."""
