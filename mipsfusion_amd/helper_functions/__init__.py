"""Host-side counterparts of the reference's ``helper_functions`` used on the hot path."""
