// worker.h -- one background thread + a message queue polled from Python: the shape shared by the
// reference's *Thread wrappers (cpp/opticalflow_thread.h:81-205, cpp/tracker_thread.h:19-100,
// cpp/refiner_thread.h).  Protocol: the worker pushes progress / result messages, then an optional
// error, then `true`; the owner polls TryPop()/Empty(), may RequestStop(), and Join()s.
#pragma once

#include <deque>
#include <functional>
#include <mutex>
#include <optional>
#include <string>
#include <thread>

// The reference stores std::unique_ptr<std::exception> made from a sliced copy, so Python sees
// what() == "std::exception"; the real message is kept here.
struct CppException {
    std::string message;
    const char* what() const { return message.c_str(); }
};

template <typename Message>
class Worker {
   public:
    Worker() = default;
    Worker(const Worker&) = delete;
    Worker& operator=(const Worker&) = delete;
    virtual ~Worker() { Join(); }

    void Join() {
        if (thread_.joinable()) thread_.join();
    }
    std::optional<Message> TryPop() {
        std::lock_guard<std::mutex> lk(queue_mtx_);
        if (queue_.empty()) return std::nullopt;
        Message m = std::move(queue_.front());
        queue_.pop_front();
        return m;
    }
    bool Empty() const {
        std::lock_guard<std::mutex> lk(queue_mtx_);
        return queue_.empty();
    }

   protected:
    void Push(Message m) {
        std::lock_guard<std::mutex> lk(queue_mtx_);
        queue_.push_back(std::move(m));
    }
    // Runs `body` on the worker thread; exceptions become a CppException message (or whatever
    // `on_error` turns them into), and `true` is always the last message.
    void Start(std::function<void()> body, std::function<void(const std::string&)> on_error = nullptr) {
        thread_ = std::thread([this, body = std::move(body), on_error = std::move(on_error)] {
            try {
                body();
            } catch (const std::exception& e) {
                if (on_error) on_error(e.what());
                else Push(CppException{e.what()});
            } catch (...) {
                Push(CppException{"Unknown exception type. This should never happen!"});
            }
            Push(true);
        });
    }

   private:
    mutable std::mutex queue_mtx_;
    std::deque<Message> queue_;
    std::thread thread_;
};
