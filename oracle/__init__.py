"""CPU restatement of the reference algorithm: TEST INFRASTRUCTURE only (tests/, smoke(), bench.py cpu_baseline)."""
