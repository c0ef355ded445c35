"""Study scripts (not tests): CPU experiments that import the oracle, kept beside the tests because only test infrastructure may."""
