"""Drop-in counterparts of the reference's `fossen` package (thruster / wrench / quaternion models)."""
