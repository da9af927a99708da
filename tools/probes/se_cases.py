CASES = [("abs", dict(prior_type="gauss_bernoulli", output_type="abs", prior_rho=0.5, prior_mean=0), [0.4, 0.8, 1.2]),
         ("relu", dict(prior_type="gauss_bernoulli", output_type="relu", prior_rho=0.5, prior_mean=0), [0.4, 0.8, 1.5]),
         ("sgn", dict(prior_type="binary", output_type="sgn", prior_p_pos=0.5), [0.5, 1.5, 2.5]),
         ("door", dict(prior_type="binary", output_type="door", output_width=1.0, prior_p_pos=0.51), [1.5, 2.0, 2.5, 3.0]),
         ("modulus", dict(prior_type="gauss_bernoulli", output_type="modulus", prior_rho=0.6, prior_mean=0), [0.8, 1.5, 2.5])]
