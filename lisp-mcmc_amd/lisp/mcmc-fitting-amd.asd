;;;; mcmc-fitting-amd.asd -- the Common Lisp (SBCL) host side of libmhx.
;;;; Same exported names as the reference's :mcmc-fitting system for the
;;;; walker-adaptive-steps path; everything numeric happens below the C ABI
;;;; (include/mhx.h) in hand-written gfx950 kernels.
(asdf:defsystem #:mcmc-fitting-amd
  :description "MI355X-native drop-in for the walker-adaptive-steps path of afranson/Lisp-MCMC"
  :version "0.2.0"
  :license "MIT"
  :depends-on (#:cffi)
  :serial t
  :components ((:file "package")
               (:file "bindings")
               (:file "models")
               (:file "expr")
               (:file "walker")))
