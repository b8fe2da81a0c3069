"""ctypes bindings of the parity checkers (the real reference behind
oracle/_ref, the plain-C restatement oracle/liboracle.so): they live with the
checkers in oracle/bind.py; the tests keep importing them from here."""
from oracle.bind import *  # noqa: F401,F403
from oracle.bind import _DT, _params_ptr, _avir_call, _lancir_call  # noqa: F401
