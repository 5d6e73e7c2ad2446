"""alan_amd: MI355X-native implementation of alan's tensorised marginal-likelihood hot path."""
