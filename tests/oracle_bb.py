"""The oracle compiled for the reference's BabyBear / Poseidon2 configuration (oracle/libms_oracle_bb.so): the code of
tests/oracle.py executed under this module's name, which makes it bind the other library. Test infrastructure only."""
import os

_src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle.py")
exec(compile(open(_src).read(), _src, "exec"))
