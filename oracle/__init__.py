"""oracle/ -- CPU restatement of the reference algorithm.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  PARITY UNPINNED upstream (the reference has no tests / golden vectors and its
TensorFlow dependency cannot be installed here); see iaf_oracle.py for the pins.
"""
