"""Drop-in counterpart of the reference's `Koopman` package."""
