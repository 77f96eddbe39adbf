"""Option parsing of the daccord-compatible front end (flag and value joined, options before positionals:
src/daccord.cpp:185-207, README.md:99)."""
import pytest
from daccord_amd import cli


def test_defaults_match_the_reference():
    opt, pos = cli.parse_args(["a.las", "a.db"])
    assert (opt["w"], opt["a"], opt["m"], opt["D"], opt["k"], opt["minfilterfreq"], opt["maxfilterfreq"], opt["l"]) == (40, 10, 3, 5000, "8", 0, 2, 0)
    assert pos == ["a.las", "a.db"] and opt["f"] is False and opt["d"] is None and opt["e"] is None


def test_joined_values_and_intervals():
    opt, pos = cli.parse_args(["-w48", "-a12", "-k10,12", "-f", "-d30", "-e40", "-l500", "-D100", "-I5,20", "-J1,8", "-t16",
                               "--minfilterfreq1", "--maxfilterfreq3", "x.las", "x.db", "y.db"])
    assert (opt["w"], opt["a"], opt["k"], opt["f"], opt["d"], opt["e"], opt["l"], opt["D"]) == (48, 12, "10,12", True, 30, 40, 500, 100)
    assert (opt["I"], opt["J"], opt["minfilterfreq"], opt["maxfilterfreq"]) == ("5,20", "1,8", 1, 3)
    assert pos == ["x.las", "x.db", "y.db"]


def test_error_profile_sources(tmp_path):
    opt, _ = cli.parse_args(["--eprof0.12,0.02,0.85", "a.las", "a.db"])
    assert cli.load_eprof(opt, "a.las") == [0.12, 0.02, 0.85]
    f = tmp_path / "x.las.eprof"; f.write_text("0.1 0.03 0.8\n")
    opt, _ = cli.parse_args(["a.las", "a.db"])
    assert cli.load_eprof(opt, str(tmp_path / "x.las")) == [0.1, 0.03, 0.8]
    with pytest.raises(SystemExit):
        cli.load_eprof(opt, str(tmp_path / "missing.las"))


def test_unknown_and_estimator_options_fail_loudly():
    for bad in (["-Q3", "a.las", "a.db"], ["--eprofonly", "a.las", "a.db"], ["a.las"]):
        with pytest.raises(SystemExit):
            cli.parse_args(bad)
