"""SURVEY section 5 "race detection / sanitizers": the emulated kernel suites under AddressSanitizer + UndefinedBehaviorSanitizer
(tests/run_sanitized.sh builds the hipemu test library with -fsanitize=address,undefined and preloads the runtimes).  51 minutes on 8 cores,
so it only runs on request (DISCO_RUN_SLOW=1 python -m pytest tests/test_sanitized_cpu.py); the last full run is profiles/r05_sanitizer_run.log."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get('DISCO_RUN_SLOW') != '1', reason='sanitizer leg: ~50 minutes; set DISCO_RUN_SLOW=1 (last run: profiles/r05_sanitizer_run.log)')
def test_emulated_suites_under_asan_ubsan():
    p = subprocess.run(['bash', os.path.join(HERE, 'run_sanitized.sh')], capture_output=True, text=True)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0 and 'ERROR: AddressSanitizer' not in tail and 'runtime error' not in tail, tail


def test_sanitizer_script_is_what_the_log_says():
    """The committed log names the flags of the script: a reader who re-runs it gets the same leg."""
    sh = open(os.path.join(HERE, 'run_sanitized.sh')).read()
    log = open(os.path.join(os.path.dirname(HERE), 'profiles', 'r05_sanitizer_run.log')).read()
    assert '-fsanitize=address,undefined' in sh and '-fsanitize=address,undefined' in log and 'passed' in log
