"""Mirror of the reference's ``inference`` package: the on-device parts of the Jacobian-field visualisation."""
